// GpuMixtureScorer.hpp -- the binding a maintainer of kkromberg/SpeechRecognition would drop into
// src/sietill/ to put libsrgpu.so behind the reference's own FeatureScorer / Recognizer interfaces.
// It is written against the REFERENCE's headers (FeatureScorer.hpp, Iter.hpp, Lexicon.hpp, TdpModel.hpp,
// Corpus.hpp), so it only compiles inside that tree: tests/test_integration_stub.py syntax-checks it
// against /root/reference when that is present.  Nothing in this repo's product path includes it.
//
//   GpuMixtureScorer : FeatureScorer     plug-in for `"feature-scorer": "gmm-gpu"` in SieTill.cpp:116-131;
//                                        per-sequence dense table like NeuralNetwork (NeuralNetwork.cpp:184-199)
//   gpu_recognize(...)                   batch replacement for the utterance loop of Recognizer::recognize
//                                        (Recognizer.cpp:46-78): one device pass over the whole Corpus, or -- given
//                                        several scorer replicas -- one shard per device
#ifndef __GPU_MIXTURE_SCORER_HPP__
#define __GPU_MIXTURE_SCORER_HPP__

#include <atomic>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "Corpus.hpp"
#include "FeatureScorer.hpp"
#include "Lexicon.hpp"
#include "TdpModel.hpp"
#include "srgpu.h"

class GpuMixtureScorer : public FeatureScorer {
public:
  // same arguments as MixtureModel's constructor in recognize mode (Mixtures.cpp:156-174); pooling is
  // MixtureModel::VarianceModel cast to int
  GpuMixtureScorer(std::string const& mixture_path, size_t dimension, int pooling, bool max_approx, int device = 0)
                  : dimension_(dimension), model_(NULL), id_(new_id()) {
    if (sr_model_load_mixset(mixture_path.c_str(), dimension, pooling, max_approx, device, &model_) != SR_OK) {
      throw std::runtime_error(sr_last_error());
    }
    uint32_t d; uint64_t c;
    sr_model_info(model_, &d, &num_states_, &c);
  }
  virtual ~GpuMixtureScorer() { sr_model_destroy(model_); }

  // Safe under the `#pragma omp parallel for` of Recognizer::recognize (Recognizer.cpp:46), which calls
  // prepare_sequence + score for a different segment on every thread through ONE scorer object
  // (Recognizer.cpp:104): the per-sequence table lives in a per-thread slot (NeuralNetwork keeps its
  // table in the object and is not safe there), and the device call is serialised -- srgpu.h: a handle
  // is used by one host thread at a time.  The threads then overlap one segment's device scoring with
  // the others' host-side search; gpu_recognize() below is still the faster route for whole corpora.
  virtual void prepare_sequence(FeatureIter const& start, FeatureIter const& end) {
    Sequence& seq = sequence();
    const size_t n_frames = end - start;
    seq.start = *start;
    seq.table.resize(n_frames * num_states_);
    std::lock_guard<std::mutex> lock(device_mutex_);
    if (sr_score_frames(model_, *start, n_frames, SR_GMM_DEFAULT, seq.table.data()) != SR_OK) {
      throw std::runtime_error(sr_last_error());
    }
  }

  virtual double score(FeatureIter const& iter, StateIdx state_idx) const {
    Sequence const& seq = sequence();
    const size_t frame = (*iter - seq.start) / dimension_;  // as NeuralNetwork::score recovers it (NeuralNetwork.cpp:196-198)
    return seq.table[frame * num_states_ + state_idx];
  }

  sr_model* handle() const { return model_; }

private:
  struct Sequence {
    const float*        start;
    std::vector<double> table;
    Sequence() : start(NULL) {}
  };
  // The calling thread's slot for THIS scorer.  The slots live in the object (one per thread that ever called
  // prepare_sequence on it) and go with it; a thread caches the slot of the scorer it used last, keyed by the scorer's
  // unique id -- not its address: a scorer constructed where a destroyed one stood must not inherit that one's table
  // (other num_states_ = wrong stride) -- so score(), called once per (frame, state) by the reference's search, takes no
  // lock.  score() must run on the thread that ran prepare_sequence for the sequence (Recognizer.cpp:104-179 does).
  Sequence& sequence() const {
    static thread_local uint64_t  last_id = 0;  // (ids start at 1)
    static thread_local Sequence* last    = NULL;
    if (last_id != id_) {
      std::lock_guard<std::mutex> lock(slots_mutex_);
      last    = &slots_[std::this_thread::get_id()];  // (map nodes do not move when others are inserted)
      last_id = id_;
    }
    return *last;
  }
  static uint64_t new_id() {
    static std::atomic<uint64_t> next(0);
    return ++next;
  }

  size_t             dimension_;
  sr_model*          model_;
  uint32_t           num_states_;
  uint64_t           id_;
  mutable std::mutex device_mutex_;
  mutable std::mutex slots_mutex_;
  mutable std::map<std::thread::id, Sequence> slots_;
};

// Whole-corpus recognition on the device: what Recognizer::recognize's loop body computes per segment
// (Recognizer.cpp:48-56), for every segment at once.  Fills `recognized` with one word sequence per segment.
inline void gpu_recognize(GpuMixtureScorer const& scorer, Lexicon const& lexicon, TdpModel const& tdp_model,
                          double tdp_loop, double tdp_forward, double tdp_skip,
                          double am_threshold, double word_penalty, Corpus const& corpus, size_t corpus_size,
                          std::vector<std::vector<WordIdx> >& recognized) {
  std::vector<uint32_t> word_off(1, 0u);
  std::vector<uint16_t> automaton;
  for (WordIdx w = 0; w < lexicon.num_words(); w++) {
    MarkovAutomaton const& a = lexicon.get_automaton_for_word(w);
    automaton.insert(automaton.end(), a.states.begin(), a.states.end());
    word_off.push_back(automaton.size());
  }
  const double tdp[3] = {tdp_loop, tdp_forward, tdp_skip};  // TdpModel keeps them private (TdpModel.hpp:25-28)
  sr_lexicon* net = NULL;
  if (sr_lexicon_create(scorer.handle(), lexicon.num_words(), word_off.data(), automaton.data(), lexicon.silence_idx(),
                        tdp, tdp_model.silence_state, &net) != SR_OK) {
    throw std::runtime_error(sr_last_error());
  }
  // Corpus keeps FLOAT offsets (Corpus.cpp:104); the ABI wants frame offsets
  const size_t dim = corpus.get_features_per_timeframe();
  std::vector<uint64_t> frame_off(corpus_size + 1, 0u);
  for (size_t s = 0; s < corpus_size; s++) {
    frame_off[s + 1] = corpus.get_feature_offsets(s).second / dim;
  }
  std::vector<uint32_t> words(frame_off[corpus_size] + 1);
  std::vector<uint64_t> out_off(corpus_size + 1);
  sr_search_params p = sr_search_params();  // zeroed, then field by field
  p.am_threshold = am_threshold;
  p.word_penalty = word_penalty;
  p.gmm_kernel = SR_GMM_DEFAULT;
  const int rc = sr_recognize_batch(scorer.handle(), net, &p, *corpus.get_feature_sequence(0).first, frame_off.data(),
                                    corpus_size, words.data(), out_off.data());
  sr_lexicon_destroy(net);
  if (rc != SR_OK) {
    throw std::runtime_error(sr_last_error());
  }
  recognized.resize(corpus_size);
  for (size_t s = 0; s < corpus_size; s++) {
    recognized[s].assign(words.begin() + out_off[s], words.begin() + out_off[s + 1]);
  }
}

// The same over several devices: Recognizer::recognize's `#pragma omp parallel for` over segments (Recognizer.cpp:46-47) at
// device granularity.  One GpuMixtureScorer replica per entry of `scorers` (each constructed on its device from the same
// model file); the library deals the segments to them by frames (greedy LPT), runs one host thread per replica and returns
// the word sequences in corpus order.  No collective: segments are independent.
inline void gpu_recognize(std::vector<GpuMixtureScorer*> const& scorers, Lexicon const& lexicon, TdpModel const& tdp_model,
                          double tdp_loop, double tdp_forward, double tdp_skip,
                          double am_threshold, double word_penalty, Corpus const& corpus, size_t corpus_size,
                          std::vector<std::vector<WordIdx> >& recognized) {
  std::vector<uint32_t> word_off(1, 0u);
  std::vector<uint16_t> automaton;
  for (WordIdx w = 0; w < lexicon.num_words(); w++) {
    MarkovAutomaton const& a = lexicon.get_automaton_for_word(w);
    automaton.insert(automaton.end(), a.states.begin(), a.states.end());
    word_off.push_back(automaton.size());
  }
  const double tdp[3] = {tdp_loop, tdp_forward, tdp_skip};
  std::vector<sr_model*>   models;
  std::vector<sr_lexicon*> nets;
  struct Cleanup {
    std::vector<sr_lexicon*>& n;
    ~Cleanup() { for (size_t i = 0; i < n.size(); i++) sr_lexicon_destroy(n[i]); }
  } cleanup = {nets};
  for (size_t d = 0; d < scorers.size(); d++) {
    sr_lexicon* net = NULL;
    if (sr_lexicon_create(scorers[d]->handle(), lexicon.num_words(), word_off.data(), automaton.data(), lexicon.silence_idx(),
                          tdp, tdp_model.silence_state, &net) != SR_OK) {
      throw std::runtime_error(sr_last_error());
    }
    models.push_back(scorers[d]->handle());
    nets.push_back(net);
  }
  const size_t dim = corpus.get_features_per_timeframe();
  std::vector<uint64_t> frame_off(corpus_size + 1, 0u);
  for (size_t s = 0; s < corpus_size; s++) {
    frame_off[s + 1] = corpus.get_feature_offsets(s).second / dim;
  }
  std::vector<uint32_t> words(frame_off[corpus_size] + 1);
  std::vector<uint64_t> out_off(corpus_size + 1);
  sr_search_params p = sr_search_params();  // zeroed, then field by field
  p.am_threshold = am_threshold;
  p.word_penalty = word_penalty;
  p.gmm_kernel = SR_GMM_DEFAULT;
  if (sr_recognize_batch_multi(models.data(), nets.data(), models.size(), &p, *corpus.get_feature_sequence(0).first,
                               frame_off.data(), corpus_size, words.data(), out_off.data(), NULL) != SR_OK) {
    throw std::runtime_error(sr_last_error());
  }
  recognized.resize(corpus_size);
  for (size_t s = 0; s < corpus_size; s++) {
    recognized[s].assign(words.begin() + out_off[s], words.begin() + out_off[s + 1]);
  }
}

#endif /* __GPU_MIXTURE_SCORER_HPP__ */
