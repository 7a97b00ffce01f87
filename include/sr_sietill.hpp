// sr_sietill.hpp -- C++ host-side mirror of the reference's scorer / search interface, on top of the
// C ABI in srgpu.h.  Header-only; link with libsrgpu.so.
//
// Same names, argument meaning and error behaviour as the reference classes for this path, so code
// (and tests) written against `sietill` read the same:
//
//   sr::FeatureScorer          FeatureScorer                sietill/FeatureScorer.hpp:12-16
//   sr::Lexicon                Lexicon / MarkovAutomaton    sietill/Lexicon.hpp:16-33, MarkovAutomaton.hpp:17-67
//   sr::TdpModel               TdpModel                     sietill/TdpModel.hpp:13-29, TdpModel.cpp:19-29
//   sr::MixtureModel           MixtureModel (as scorer)     sietill/Mixtures.hpp:18-92
//   sr::Corpus                 Corpus (feature store)       sietill/Corpus.hpp:55-84
//   sr::Recognizer             Recognizer                   sietill/Recognizer.hpp:91-132
//   sr::Aligner                Aligner                      sietill/Alignment.hpp:19-63
//   sr::FeaturePostProcessor   SignalAnalysis::process_features  sietill/SignalAnalysis.cpp:320-336,340-349,379-399
//   sr::read_feature_file, write_alignment, read_alignment    sietill/IO.cpp:48-69, Alignment.cpp:303-318
//   sr::Trainer (re-alignment) Trainer::train's align loop   sietill/Training.cpp:163-184, :239-253, :585-612
//
// Differences, all forced by the device boundary: features are passed as (pointer, frame count)
// instead of FeatureIter pairs; MixtureModel::prepare_sequence really does work (it fills the dense
// score table on the GPU, like NeuralNetwork::prepare_sequence does on the CPU,
// sietill/NeuralNetwork.cpp:184-199); errors that the reference reports with abort()
// (Mixtures.cpp:97-102) are thrown as std::runtime_error carrying the same text.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <time.h>
#include <string>
#include <utility>
#include <vector>

#include "srgpu.h"

namespace sr {

typedef size_t WordIdx;     // sietill/Types.hpp:14-17
typedef uint16_t StateIdx;

inline void check(int rc) {
  if (rc != SR_OK) throw std::runtime_error(sr_last_error());
}
// the loaded library must lay its parameter structs out like the header this file was compiled against (srgpu.h)
inline void check_abi() {
  static const bool ok = sr_abi_version() == SR_ABI_VERSION;
  if (!ok) throw std::runtime_error("libsrgpu.so: ABI version differs from the srgpu.h this binding was compiled against");
}

// ---- FeatureScorer.hpp:12-16 ----------------------------------------------------------------------
class FeatureScorer {
 public:
  virtual ~FeatureScorer() {}
  virtual void prepare_sequence(const float* begin, size_t n_frames) = 0;
  virtual double score(size_t frame, StateIdx state_idx) const = 0;
};

// ---- MarkovAutomaton.hpp:17-67, Lexicon.cpp:11-62 ---------------------------------------------------
struct MarkovAutomaton {
  std::vector<StateIdx> states;
  MarkovAutomaton() {}
  MarkovAutomaton(StateIdx start, uint16_t num, uint16_t repetitions) {
    for (StateIdx s = start; s < start + num; s++) states.insert(states.end(), repetitions, s);
  }
  StateIdx first_state() const { return states.front(); }
  StateIdx last_state() const { return states.back(); }
  size_t num_states() const { return states.size(); }
  StateIdx operator[](size_t i) const { return states[i]; }
  static MarkovAutomaton concat(std::vector<MarkovAutomaton const*> automata) {
    MarkovAutomaton r;
    for (auto a : automata) r.states.insert(r.states.end(), a->states.begin(), a->states.end());
    return r;
  }
};

class Lexicon {
 public:
  WordIdx add_word(std::string const& orth, uint16_t num_states, uint16_t state_repetitions, bool silence = false) {
    const WordIdx w = automata_.size();
    if (silence) silence_ = w;
    const StateIdx start = automata_.empty() ? 0 : StateIdx(automata_.back().last_state() + 1);
    orth_.push_back(orth);
    automata_.push_back(MarkovAutomaton(start, num_states, state_repetitions));
    return w;
  }
  MarkovAutomaton const& get_silence_automaton() const { return automata_[silence_]; }
  MarkovAutomaton const& get_automaton_for_word(WordIdx w) const { return automata_[w]; }
  StateIdx num_states() const { return automata_.back().last_state() + 1; }
  WordIdx num_words() const { return automata_.size(); }
  WordIdx silence_idx() const { return silence_; }
  WordIdx operator[](std::string const& orth) const {
    auto it = std::find(orth_.begin(), orth_.end(), orth);
    return it == orth_.end() ? WordIdx(-1) : WordIdx(it - orth_.begin());  // Lexicon.cpp:57-62
  }

 private:
  std::vector<std::string> orth_;
  std::vector<MarkovAutomaton> automata_;
  WordIdx silence_ = 0;
};

// ---- TdpModel.cpp:19-29 ------------------------------------------------------------------------------
struct TdpModel {
  StateIdx silence_state;
  double tdp_loop, tdp_forward, tdp_skip;
  TdpModel(StateIdx silence_state, double loop, double forward, double skip)
      : silence_state(silence_state), tdp_loop(loop), tdp_forward(forward), tdp_skip(skip) {}
  double score(StateIdx to, size_t jump) const {
    if (to == silence_state) return tdp_forward;
    switch (jump) {
      case 0: return tdp_loop;
      case 1: return tdp_forward;
      case 2: return tdp_skip;
    }
    return std::numeric_limits<double>::infinity();
  }
};

// ---- Mixtures.hpp:18-92: the GMM as a FeatureScorer, resident on one GPU ----------------------------------
class MixtureModel : public FeatureScorer {
 public:
  enum VarianceModel { GLOBAL_POOLING, MIXTURE_POOLING, NO_POOLING };  // Mixtures.hpp:20-24

  // MixtureModel(config, dimension, num_mixtures, var_model, max_approx) with action "recognize" and
  // "load-mixtures-from" = path (Mixtures.cpp:156-174)
  MixtureModel(std::string const& load_mixtures_from, size_t dimension, VarianceModel var_model, bool max_approx,
               int device = 0, int gmm_kernel = SR_GMM_DEFAULT)
      : dimension(dimension), var_model(var_model), gmm_kernel(gmm_kernel), path_(load_mixtures_from), max_approx_(max_approx),
        device_(device) {
    check_abi();
    check(sr_model_load_mixset(load_mixtures_from.c_str(), (uint32_t)dimension, (int)var_model, max_approx ? 1 : 0, device, &h_));
    uint32_t d, s;
    uint64_t c;
    check(sr_model_info(h_, &d, &s, &c));
    num_mixtures_ = s;
    num_densities_ = c;
  }
  ~MixtureModel() { sr_model_destroy(h_); }
  MixtureModel(MixtureModel const&) = delete;
  MixtureModel& operator=(MixtureModel const&) = delete;

  const size_t dimension;
  const VarianceModel var_model;
  int gmm_kernel;

  size_t num_mixtures() const { return num_mixtures_; }
  size_t num_densities() const { return num_densities_; }
  sr_model* handle() const { return h_; }
  int device() const { return device_; }
  // another replica of the same model file on `device` (utterance batches shard across devices, every device holds the
  // whole model: Recognizer::recognize(corpus, devices))
  std::unique_ptr<MixtureModel> replicate(int device) const {
    return std::unique_ptr<MixtureModel>(new MixtureModel(path_, dimension, var_model, max_approx_, device, gmm_kernel));
  }

  // FeatureScorer: one dense [T x S] table per sequence
  void prepare_sequence(const float* begin, size_t n_frames) override {
    table_.resize(n_frames * num_mixtures_);
    check(sr_score_frames(h_, begin, n_frames, gmm_kernel, table_.data()));
  }
  double score(size_t frame, StateIdx mixture_idx) const override { return table_[frame * num_mixtures_ + mixture_idx]; }

 private:
  sr_model* h_ = nullptr;
  size_t num_mixtures_ = 0, num_densities_ = 0;
  std::vector<double> table_;
  std::string path_;
  bool max_approx_ = true;
  int device_ = 0;
};

// ---- Corpus.hpp:55-84: contiguous features + offsets + reference word sequences ---------------------------
class Corpus {
 public:
  explicit Corpus(size_t features_per_timeframe, double frame_duration = 0.010)
      : features_per_timeframe_(features_per_timeframe), frame_duration_(frame_duration) {
    frame_offsets_.push_back(0);
    orth_offsets_.push_back(0);
  }
  void add_segment(const float* feats, size_t n_frames, std::vector<WordIdx> const& orth) {
    features_.insert(features_.end(), feats, feats + n_frames * features_per_timeframe_);
    frame_offsets_.push_back(frame_offsets_.back() + n_frames);
    orths_.insert(orths_.end(), orth.begin(), orth.end());
    orth_offsets_.push_back(orths_.size());
  }
  size_t get_corpus_size() const { return orth_offsets_.size() - 1; }
  size_t get_total_frame_count() const { return frame_offsets_.back(); }
  size_t get_features_per_timeframe() const { return features_per_timeframe_; }
  double get_frame_duration() const { return frame_duration_; }
  std::pair<const WordIdx*, const WordIdx*> get_word_sequence(size_t s) const {
    return {orths_.data() + orth_offsets_[s], orths_.data() + orth_offsets_[s + 1]};
  }
  std::pair<const float*, size_t> get_feature_sequence(size_t s) const {
    return {features_.data() + frame_offsets_[s] * features_per_timeframe_, (size_t)(frame_offsets_[s + 1] - frame_offsets_[s])};
  }
  const float* features() const { return features_.data(); }
  const uint64_t* frame_offsets() const { return frame_offsets_.data(); }

 private:
  size_t features_per_timeframe_;
  double frame_duration_;
  std::vector<uint64_t> frame_offsets_;
  std::vector<float> features_;
  std::vector<size_t> orth_offsets_;
  std::vector<WordIdx> orths_;
};

// ---- Recognizer.hpp:18-47 ---------------------------------------------------------------------------------
struct EDAccumulator {
  uint16_t total_count = 0, substitute_count = 0, insert_count = 0, delete_count = 0;
  EDAccumulator& operator+=(EDAccumulator const& o) {
    total_count += o.total_count; substitute_count += o.substitute_count;
    insert_count += o.insert_count; delete_count += o.delete_count;
    return *this;
  }
  void substitution_error() { total_count++; substitute_count++; }
  void insertion_error() { total_count++; insert_count++; }
  void deletion_error() { total_count++; delete_count++; }
};

struct RecognitionStats {  // what Recognizer::recognize prints (Recognizer.cpp:82-91)
  EDAccumulator errors;
  size_t ref_words = 0, sentence_errors = 0, corpus_size = 0;
  double wer = 0, ser = 0, seconds = 0, rtf = 0;
  std::vector<std::vector<WordIdx>> hypotheses;
  std::vector<uint64_t> frames_per_device;  // recognize(corpus, devices): how the LPT deal loaded the devices
};

class Recognizer {
 public:
  // Recognizer(config, lexicon, scorer, tdp_model): "am-threshold" (20.0), "word-penalty" (10.0),
  // "max-recognition-runs" (1000) as in Recognizer.cpp:31-34
  Recognizer(Lexicon const& lexicon, MixtureModel& scorer, TdpModel const& tdp_model, double am_threshold = 20.0,
             double word_penalty = 10.0, size_t max_recognition_runs = 1000)
      : am_threshold_(am_threshold), word_penalty_(word_penalty), max_recognition_runs_(max_recognition_runs),
        lexicon_(lexicon), scorer_(scorer) {
    word_off_.assign(1, 0);
    for (WordIdx w = 0; w < lexicon.num_words(); w++) {
      auto const& a = lexicon.get_automaton_for_word(w);
      automaton_.insert(automaton_.end(), a.states.begin(), a.states.end());
      word_off_.push_back((uint32_t)automaton_.size());
    }
    tdp_[0] = tdp_model.tdp_loop; tdp_[1] = tdp_model.tdp_forward; tdp_[2] = tdp_model.tdp_skip;
    silence_state_ = tdp_model.silence_state;
    net_ = make_net(scorer);
  }
  ~Recognizer() {
    for (auto& r : replicas_) sr_lexicon_destroy(r.net);
    sr_lexicon_destroy(net_);
  }
  Recognizer(Recognizer const&) = delete;

  sr_search_params search_params() const {
    sr_search_params p = sr_search_params();  // zeroed, then field by field: a field added to the struct cannot shift these
    p.am_threshold = am_threshold_;
    p.word_penalty = word_penalty_;
    p.gmm_kernel = scorer_.gmm_kernel;
    return p;
  }

  // Recognizer::recognizeSequence_pruned (Recognizer.cpp:103-232)
  void recognizeSequence_pruned(const float* feature_begin, size_t n_frames, std::vector<WordIdx>& output) {
    const uint64_t off[2] = {0, n_frames};
    std::vector<uint32_t> words(std::max<size_t>(n_frames, 1));
    uint64_t woff[2];
    const sr_search_params p = search_params();
    check(sr_recognize_batch(scorer_.handle(), net_, &p, feature_begin, off, 1, words.data(), woff));
    output.assign(words.begin(), words.begin() + woff[1]);
  }

  // Recognizer::recognize (Recognizer.cpp:38-92): the whole corpus in one device pass.
  // With `devices`: the reference's `#pragma omp parallel for` over segments (:46-47) at device granularity -- the
  // segments are dealt to the devices by frames (greedy LPT), every device gets a replica of the model and of the search
  // network (created on first use, kept) and one host thread that feeds and recognises its shard; hypotheses come back in
  // corpus order.  A device may be listed more than once (two replicas on it).  No collective.
  RecognitionStats recognize(Corpus const& corpus) { return recognize(corpus, std::vector<int>()); }
  RecognitionStats recognize(Corpus const& corpus, std::vector<int> const& devices) {
    RecognitionStats st;
    const size_t n = std::min(corpus.get_corpus_size(), max_recognition_runs_);
    st.corpus_size = n;
    const uint64_t total = corpus.frame_offsets()[n];
    std::vector<uint32_t> words(std::max<uint64_t>(total, 1));
    std::vector<uint64_t> woff(n + 1);
    const sr_search_params p = search_params();
    double t0;
    if (devices.size() <= 1 && (devices.empty() || devices[0] == scorer_.device())) {
      t0 = now();
      check(sr_recognize_batch(scorer_.handle(), net_, &p, corpus.features(), corpus.frame_offsets(), (uint32_t)n,
                               words.data(), woff.data()));
    } else {
      std::vector<sr_model*> models;
      std::vector<sr_lexicon*> nets;
      std::vector<size_t> used(replicas_.size(), 0);
      bool own_used = false;
      for (int dev : devices) {  // the recogniser's own model serves its device once, replicas the rest
        if (dev == scorer_.device() && !own_used) { own_used = true; models.push_back(scorer_.handle()); nets.push_back(net_); continue; }
        size_t r = 0;
        for (; r < replicas_.size(); r++)
          if (replicas_[r].model->device() == dev && !used[r]) break;
        if (r == replicas_.size()) {
          Replica rep;
          rep.model = scorer_.replicate(dev);
          rep.net = make_net(*rep.model);
          replicas_.push_back(std::move(rep));
          used.push_back(0);
        }
        used[r] = 1;
        models.push_back(replicas_[r].model->handle());
        nets.push_back(replicas_[r].net);
      }
      st.frames_per_device.assign(devices.size(), 0);
      t0 = now();
      check(sr_recognize_batch_multi(models.data(), nets.data(), (uint32_t)models.size(), &p, corpus.features(),
                                     corpus.frame_offsets(), (uint32_t)n, words.data(), woff.data(), st.frames_per_device.data()));
    }
    st.seconds = now() - t0;
    for (size_t s = 0; s < n; s++) {
      std::vector<WordIdx> hyp(words.begin() + woff[s], words.begin() + woff[s + 1]);
      auto ref = corpus.get_word_sequence(s);
      EDAccumulator ed = editDistance(ref.first, ref.second, hyp.data(), hyp.data() + hyp.size());
      st.errors += ed;
      st.ref_words += ref.second - ref.first;
      if (ed.total_count > 0) st.sentence_errors++;
      st.hypotheses.push_back(std::move(hyp));
    }
    st.wer = 100.0 * st.errors.total_count / (double)st.ref_words;
    st.ser = 100.0 * st.sentence_errors / (double)n;
    st.rtf = st.seconds / (corpus.get_frame_duration() * (double)total);  // Recognizer.cpp:85
    return st;
  }

  // Recognizer::editDistance (Recognizer.cpp:332-389), including its 16-bit counters and the stale row-0
  // insertion counter (`current_rates[0].insertion_error()` after the swap, :349-352)
  static EDAccumulator editDistance(const WordIdx* ref_begin, const WordIdx* ref_end, const WordIdx* rec_begin,
                                    const WordIdx* rec_end) {
    const size_t ref_size = ref_end - ref_begin, hyp_size = rec_end - rec_begin;
    std::vector<EDAccumulator> current(1 + ref_size), previous(1 + ref_size);
    for (size_t r = 1; r <= ref_size; r++) { current[r] = current[r - 1]; current[r].deletion_error(); }
    for (size_t h = 1; h <= hyp_size; h++) {
      current.swap(previous);
      current[0].insertion_error();
      for (size_t r = 1; r <= ref_size; r++) {
        uint16_t best = 0xFFFF;
        if (previous[r - 1].total_count < best && ref_begin[r - 1] == rec_begin[h - 1]) {
          current[r] = previous[r - 1]; best = current[r].total_count;
        }
        if (previous[r - 1].total_count + 1 < best) {
          current[r] = previous[r - 1]; current[r].substitution_error(); best = current[r].total_count;
        }
        if (previous[r].total_count + 1 < best) {
          current[r] = previous[r]; current[r].insertion_error(); best = current[r].total_count;
        }
        if (current[r - 1].total_count + 1 < best) {
          current[r] = current[r - 1]; current[r].deletion_error(); best = current[r].total_count;
        }
      }
    }
    return current[ref_size];
  }

 private:
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);  // Timer.hpp:11-40
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
  }
  sr_lexicon* make_net(MixtureModel& scorer) const {
    sr_lexicon* net = nullptr;
    check(sr_lexicon_create(scorer.handle(), (uint32_t)lexicon_.num_words(), word_off_.data(), automaton_.data(),
                            (uint32_t)lexicon_.silence_idx(), tdp_, silence_state_, &net));
    return net;
  }
  struct Replica {
    std::unique_ptr<MixtureModel> model;
    sr_lexicon* net = nullptr;
  };
  const double am_threshold_, word_penalty_;
  const size_t max_recognition_runs_;
  Lexicon const& lexicon_;
  MixtureModel& scorer_;
  std::vector<uint32_t> word_off_;
  std::vector<uint16_t> automaton_;
  double tdp_[3] = {0, 0, 0};
  StateIdx silence_state_ = 0;
  sr_lexicon* net_ = nullptr;
  std::vector<Replica> replicas_;
};

// ---- Alignment.hpp:19-63 -------------------------------------------------------------------------------------
struct AlignmentItem {  // sietill/Types.hpp:29-38
  uint16_t count = 0;
  StateIdx state = 0;
  float weight = 0.0f;
};

class Aligner {
 public:
  Aligner(MixtureModel& mixtures, TdpModel const& tdp_model) : mixtures_(mixtures), tdp_(tdp_model) {}

  // Aligner::align_sequence_full (Alignment.cpp:50-144); alignment receives one item per frame
  double align_sequence_full(const float* feature_begin, size_t n_frames, MarkovAutomaton const& reference,
                             std::vector<AlignmentItem>& alignment) {
    return run(feature_begin, n_frames, reference, alignment, false, 0.0);
  }
  // Aligner::align_sequence_pruned (Alignment.cpp:149-288)
  double align_sequence_pruned(const float* feature_begin, size_t n_frames, MarkovAutomaton const& reference,
                               std::vector<AlignmentItem>& alignment, double pruning_threshold) {
    return run(feature_begin, n_frames, reference, alignment, true, pruning_threshold);
  }

 private:
  double run(const float* feats, size_t T, MarkovAutomaton const& ref, std::vector<AlignmentItem>& alignment, bool pruned,
             double thr) {
    const uint64_t foff[2] = {0, T}, aoff[2] = {0, ref.num_states()};
    const double tdp[3] = {tdp_.tdp_loop, tdp_.tdp_forward, tdp_.tdp_skip};
    sr_corpus* c = nullptr;
    check(sr_corpus_upload(mixtures_.handle(), feats, foff, 1, &c));
    std::vector<uint16_t> states(std::max<size_t>(T, 1));
    double cost = 0.0;
    const int rc = pruned ? sr_align_corpus_pruned(mixtures_.handle(), c, ref.states.data(), aoff, tdp, tdp_.silence_state, thr,
                                                   mixtures_.gmm_kernel, states.data(), &cost)
                          : sr_align_corpus(mixtures_.handle(), c, ref.states.data(), aoff, tdp, tdp_.silence_state,
                                            mixtures_.gmm_kernel, states.data(), &cost);
    sr_corpus_destroy(c);
    check(rc);
    alignment.assign(T, AlignmentItem());
    for (size_t t = 0; t < T; t++) { alignment[t].state = states[t]; alignment[t].weight = 1; alignment[t].count = 1; }
    return cost;
  }
  MixtureModel& mixtures_;
  TdpModel tdp_;
};

// ---- on-disk formats either side of the path ------------------------------------------
// `<feature-path><name>.mm2`: raw little-endian float32, n-features-file values per frame (IO.cpp:48-69)
inline std::vector<float> read_feature_file(std::string const& path) {
  std::vector<float> features;
  std::ifstream in(path.c_str(), std::ios_base::in | std::ios_base::binary);
  if (!in.good()) return features;  // the reference prints an error and returns an empty vector (IO.cpp:52-55)
  in.seekg(0, std::ios_base::end);
  const std::streamoff bytes = in.tellg();
  in.seekg(0);
  features.resize((size_t)bytes / sizeof(float));
  in.read(reinterpret_cast<char*>(features.data()), features.size() * sizeof(float));
  return features;
}

// What Corpus::read applies to every utterance (Corpus.cpp:101-102): static features -> static | delta | delta-delta,
// optional mean/variance normalisation (computed in double, stored as float) and per-utterance maximum
// normalisation of component 0 (SignalAnalysis.cpp:379-399, :320-336, :340-349).
struct FeaturePostProcessor {
  size_t n_features_in_file = 12, n_features_first = 12, n_features_second = 1, deriv_step = 3;  // SignalAnalysis.cpp:53-56
  bool energy_max_norm = true;                                                                   // :46
  std::vector<double> mean, stddev;  // empty: no mean/variance normalisation
  size_t n_features_total() const { return n_features_in_file + n_features_first + n_features_second; }

  // normalisation file: n_features_total f64 means then as many f64 standard deviations (SignalAnalysis.cpp:364-375)
  bool read_normalization_file(std::string const& path) {
    std::ifstream in(path.c_str(), std::ios_base::in | std::ios_base::binary);
    if (!in.good()) return false;
    mean.resize(n_features_total());
    stddev.resize(n_features_total());
    in.read(reinterpret_cast<char*>(mean.data()), sizeof(double) * mean.size());
    in.read(reinterpret_cast<char*>(stddev.data()), sizeof(double) * stddev.size());
    return (bool)in;
  }

  void process_features(std::vector<float>& features) const {
    const size_t nf = n_features_in_file, nt = n_features_total(), T = features.size() / nf;
    std::vector<float> seq(T * nt, 0.0f);
    for (size_t f = 0; f < T; f++) std::copy(features.begin() + f * nf, features.begin() + (f + 1) * nf, seq.begin() + f * nt);
    // An utterance of at most deriv_step frames: the reference indexes frame max(t, deriv_step) >= T and frame
    // T - 1 - deriv_step (size_t underflow) -- reads past its buffer, results undefined (found by tools/sanitize_host.py; the
    // corpus has no such utterance).  Here both windows are clamped into [0, T - 1] instead; for T > deriv_step the indices
    // below are exactly the reference's.
    for (size_t t = 0; t < T; t++) {  // first derivative, window clamped at the start (:322-328)
      const size_t a = std::min(std::max(t, deriv_step), T - 1), b = a >= deriv_step ? a - deriv_step : 0;
      for (size_t k = 0; k < n_features_first; k++) seq[t * nt + nf + k] = seq[a * nt + k] - seq[b * nt + k];
    }
    for (size_t t = 0; t < T; t++) {  // second derivative from the first, window clamped at the end (:329-335)
      const size_t a = T > deriv_step ? std::min(t, T - 1 - deriv_step) + deriv_step : T - 1;
      for (size_t k = 0; k < n_features_second; k++) seq[t * nt + nf + n_features_first + k] = seq[a * nt + nf + k] - seq[t * nt + nf + k];
    }
    if (!mean.empty()) {
      for (size_t t = 0; t < T; t++) {
        for (size_t k = 0; k < nt; k++) seq[t * nt + k] = (float)((double)seq[t * nt + k] - mean[k]);
        for (size_t k = 0; k < nt; k++) seq[t * nt + k] = (float)((double)seq[t * nt + k] / stddev[k]);
      }
    }
    if (energy_max_norm) {
      float mx = -std::numeric_limits<float>::infinity();
      for (size_t t = 0; t < T; t++) mx = std::max(mx, seq[t * nt]);
      for (size_t t = 0; t < T; t++) seq[t * nt] -= mx;
    }
    features.swap(seq);
  }
};

// alignment dump: size_t max_aligns; size_t num_frames; AlignmentItem[num_frames * max_aligns] (Alignment.cpp:303-318)
inline void write_alignment(std::ostream& out, std::vector<AlignmentItem> const& alignment, size_t max_aligns) {
  const size_t num_frames = alignment.size() / max_aligns;
  out.write(reinterpret_cast<const char*>(&max_aligns), sizeof max_aligns);
  out.write(reinterpret_cast<const char*>(&num_frames), sizeof num_frames);
  out.write(reinterpret_cast<const char*>(alignment.data()), max_aligns * num_frames * sizeof(AlignmentItem));
}
inline void read_alignment(std::istream& in, std::vector<AlignmentItem>& alignment, size_t& max_aligns) {
  size_t num_frames = 0;
  in.read(reinterpret_cast<char*>(&max_aligns), sizeof max_aligns);
  in.read(reinterpret_cast<char*>(&num_frames), sizeof num_frames);
  alignment.resize(num_frames * max_aligns);
  in.read(reinterpret_cast<char*>(alignment.data()), max_aligns * num_frames * sizeof(AlignmentItem));
}

// ---- the training-side callers of the path (Training.hpp:18-107): NOT the EM trainer, only the two loops of
// Trainer::train that run the scorer and the aligner over the whole corpus -- re-alignment (Training.cpp:163-184)
// and the average acoustic score along the alignment (calc_am_score, :585-612) -- each as one device pass.
class Trainer {
 public:
  Trainer(Lexicon const& lexicon, MixtureModel& mixtures, TdpModel const& tdp_model, double pruning_threshold = 50.0,
          bool alignment_pruning = true)
      : pruning_threshold_(pruning_threshold), alignment_pruning_(alignment_pruning), lexicon_(lexicon),
        mixtures_(mixtures), tdp_(tdp_model) {}

  // Trainer::build_segment_automaton (Training.cpp:238-253): sil w1 sil w2 ... sil
  MarkovAutomaton build_segment_automaton(const WordIdx* segment_begin, const WordIdx* segment_end) const {
    std::vector<MarkovAutomaton const*> automata;
    for (const WordIdx* it = segment_begin; it != segment_end; ++it) {
      automata.push_back(&lexicon_.get_silence_automaton());
      automata.push_back(&lexicon_.get_automaton_for_word(*it));
    }
    automata.push_back(&lexicon_.get_silence_automaton());
    return MarkovAutomaton::concat(automata);
  }

  // the re-alignment loop of Trainer::train (Training.cpp:163-184) over every segment of the corpus;
  // alignment gets one item per corpus frame, costs (optional) the per-segment path costs
  void realign(Corpus const& corpus, std::vector<AlignmentItem>& alignment, std::vector<double>* costs = nullptr) {
    const size_t n = corpus.get_corpus_size();
    std::vector<uint16_t> automata;
    std::vector<uint64_t> aut_off(1, 0);
    for (size_t s = 0; s < n; s++) {
      auto w = corpus.get_word_sequence(s);
      MarkovAutomaton a = build_segment_automaton(w.first, w.second);
      automata.insert(automata.end(), a.states.begin(), a.states.end());
      aut_off.push_back(automata.size());
    }
    const uint64_t F = corpus.get_total_frame_count();
    const double tdp[3] = {tdp_.tdp_loop, tdp_.tdp_forward, tdp_.tdp_skip};
    sr_corpus* c = nullptr;
    check(sr_corpus_upload(mixtures_.handle(), corpus.features(), corpus.frame_offsets(), (uint32_t)n, &c));
    std::vector<uint16_t> states(std::max<uint64_t>(F, 1));
    std::vector<double> cost(std::max<size_t>(n, 1));
    const int rc = alignment_pruning_
                       ? sr_align_corpus_pruned(mixtures_.handle(), c, automata.data(), aut_off.data(), tdp, tdp_.silence_state,
                                                pruning_threshold_, mixtures_.gmm_kernel, states.data(), cost.data())
                       : sr_align_corpus(mixtures_.handle(), c, automata.data(), aut_off.data(), tdp, tdp_.silence_state,
                                         mixtures_.gmm_kernel, states.data(), cost.data());
    sr_corpus_destroy(c);
    check(rc);
    alignment.assign(F, AlignmentItem());
    for (uint64_t t = 0; t < F; t++) { alignment[t].state = states[t]; alignment[t].weight = 1; alignment[t].count = 1; }
    if (costs) costs->assign(cost.begin(), cost.begin() + n);
  }

  // Trainer::calc_am_score (Training.cpp:585-612): sequential sum of score(frame, aligned state) / frames
  double calc_am_score(Corpus const& corpus, std::vector<AlignmentItem> const& alignment) {
    const uint64_t F = corpus.get_total_frame_count();
    std::vector<uint16_t> states(std::max<uint64_t>(F, 1));
    for (uint64_t t = 0; t < F; t++) states[t] = alignment[t].state;
    std::vector<double> per_frame(std::max<uint64_t>(F, 1));
    sr_corpus* c = nullptr;
    check(sr_corpus_upload(mixtures_.handle(), corpus.features(), corpus.frame_offsets(), (uint32_t)corpus.get_corpus_size(), &c));
    const int rc = sr_path_scores_corpus(mixtures_.handle(), c, states.data(), mixtures_.gmm_kernel, per_frame.data());
    sr_corpus_destroy(c);
    check(rc);
    double total_score = 0.0;
    for (uint64_t t = 0; t < F; t++) total_score += per_frame[t];
    return total_score / F;
  }

 private:
  const double pruning_threshold_;
  const bool alignment_pruning_;
  Lexicon const& lexicon_;
  MixtureModel& mixtures_;
  TdpModel tdp_;
};

// ---- Teaching::LinearSearch (rwth-asr-0.5/src/Teaching/LinearSearch.hh:9-62, SearchInterface.hh:20-30) -------------
// Bigram-LM beam search over a linear lexicon, one device pass per corpus.  The toolkit wires lexicon, language model
// and transition model through Speech::ModelCombination (LinearSearch.cc:462-475); here they are plain arrays:
//   linear_lexicon[w] = mixture (emission state) sequence of word w (LinearSearch::buildLinearLexicon :477-483),
//   lm[w * W + h]     = getLanguageModelScore(w, h) = -log p(w | h) (SearchInterface.cc:77-81),
//   tdp[isSilence][loop, forward, skip, exit] (SearchSpace::setTransitionScores :169-180).
// Parameters keep the reference's names: "acoustic-pruning", "lm-pruning" (:438-446), infinity = no beam.
class LinearSearch {
 public:
  struct TracebackItem {  // SearchInterface::TracebackItem
    uint32_t word;
    float score;
    uint32_t time;
  };
  typedef std::vector<TracebackItem> Traceback;

  LinearSearch(MixtureModel& scorer, std::vector<std::vector<uint16_t> > const& linear_lexicon, uint32_t silence,
               std::vector<float> const& lm, const float tdp[2][4], float acoustic_pruning = std::numeric_limits<float>::max(),
               float lm_pruning = std::numeric_limits<float>::max())
      : scorer_(scorer), acoustic_pruning_(acoustic_pruning), lm_pruning_(lm_pruning) {
    std::vector<uint32_t> word_off(1, 0);
    std::vector<uint16_t> mixtures;
    for (size_t w = 0; w < linear_lexicon.size(); w++) {
      mixtures.insert(mixtures.end(), linear_lexicon[w].begin(), linear_lexicon[w].end());
      word_off.push_back((uint32_t)mixtures.size());
    }
    if (lm.size() != linear_lexicon.size() * linear_lexicon.size()) throw std::runtime_error("LinearSearch: lm must be W x W");
    check(sr_bigram_create(scorer.handle(), (uint32_t)linear_lexicon.size(), word_off.data(), mixtures.data(), silence, lm.data(),
                           &tdp[0][0], &net_));
  }
  ~LinearSearch() { sr_bigram_destroy(net_); }
  LinearSearch(LinearSearch const&) = delete;
  LinearSearch& operator=(LinearSearch const&) = delete;

  // initialize() + processFrame(1..T) + getResult() (LinearSearch.cc:489-520) for every segment of the corpus
  void recognize(Corpus const& corpus, std::vector<Traceback>& results) {
    const size_t n = corpus.get_corpus_size();
    const uint64_t F = corpus.get_total_frame_count();
    std::vector<uint32_t> words(F + n + 1), times(F + n + 1);
    std::vector<float> scores(F + n + 1);
    std::vector<uint64_t> off(n + 1);
    sr_corpus* c = nullptr;
    check(sr_corpus_upload(scorer_.handle(), corpus.features(), corpus.frame_offsets(), (uint32_t)n, &c));
    sr_bigram_params p = sr_bigram_params();  // zeroed, then field by field: a field added to the struct cannot shift these
    p.acoustic_pruning = acoustic_pruning_;
    p.lm_pruning = lm_pruning_;
    p.gmm_kernel = scorer_.gmm_kernel;
    const int rc = sr_recognize_bigram_corpus(scorer_.handle(), c, net_, &p, words.data(), scores.data(), times.data(), off.data());
    sr_corpus_destroy(c);
    check(rc);
    results.assign(n, Traceback());
    for (size_t s = 0; s < n; s++)
      for (uint64_t i = off[s]; i < off[s + 1]; i++) results[s].push_back(TracebackItem{words[i], scores[i], times[i]});
  }

 private:
  MixtureModel& scorer_;
  float acoustic_pruning_, lm_pruning_;
  sr_bigram* net_ = nullptr;
};

}  // namespace sr
