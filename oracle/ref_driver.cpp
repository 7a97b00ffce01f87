/*
 * ref_driver.cpp -- C shim over the REAL reference classes (test infrastructure only).
 *
 * This file is our own glue; it #includes the reference's headers from where they lie under
 * /root/reference/src/sietill (never copied into this repo) and is linked with the reference's
 * own translation units by oracle/Makefile into oracle/_ref/libsietill_ref.so.  It exists to pin
 * oracle/sr_oracle.c against the actual reference and to generate tests/golden/ vectors
 * (oracle/gen_golden.py).  Nothing in the product path loads it.
 *
 * Reference entry points driven here:
 *   MixtureModel(config, dim, S, pooling, max_approx)  sietill/Mixtures.cpp:156-195 (read :748, finalize :374)
 *   MixtureModel::score                                sietill/Mixtures.cpp:737-744
 *   Recognizer::recognizeSequence_pruned               sietill/Recognizer.cpp:103-232
 *   Recognizer::editDistance                           sietill/Recognizer.cpp:332-389
 *   Aligner::align_sequence_full / _pruned             sietill/Alignment.cpp:50-144 / :149-288
 */
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "Alignment.hpp"
#include "Config.hpp"
#include "Lexicon.hpp"
#include "Mixtures.hpp"
#include "Corpus.hpp"
#include "Recognizer.hpp"
#include "SignalAnalysis.hpp"
#include "TdpModel.hpp"
#include "Training.hpp"

#include <fstream>

namespace {
struct RefCtx {
  std::unique_ptr<Configuration> config;
  std::unique_ptr<Lexicon> lexicon;
  std::unique_ptr<MixtureModel> mixtures;
  std::unique_ptr<TdpModel> tdp;
  uint32_t dim;
};

// Features are copied into a buffer padded by 4 floats: density_score_sse's _mm_loadu_ps reads two
// floats past the last pair of the final frame (Mixtures.cpp:653).
std::vector<float> padded(const float* feats, size_t T, uint32_t dim) {
  std::vector<float> buf(T * dim + 4, 0.0f);
  std::memcpy(buf.data(), feats, sizeof(float) * T * dim);
  return buf;
}
}  // namespace

extern "C" {

// config_path: JSON with action=recognize, load-mixtures-from, verbosity=noLog, tdp-*, am-threshold,
// word-penalty.  Lexicon: n_words entries of (num_states, repetitions); word `silence_word` is silence.
void* ref_create(const char* config_path, uint32_t dim, uint32_t n_words, const uint16_t* word_states,
                 const uint16_t* word_reps, uint32_t silence_word, int pooling, int max_approx) {
  RefCtx* c = new RefCtx();
  c->dim = dim;
  c->config.reset(new Configuration(std::string(config_path)));
  c->lexicon.reset(new Lexicon());
  for (uint32_t w = 0; w < n_words; w++) {
    c->lexicon->add_word("w" + std::to_string(w), word_states[w], word_reps[w], w == silence_word);
  }
  c->mixtures.reset(new MixtureModel(*c->config, dim, c->lexicon->num_states(),
                                     static_cast<MixtureModel::VarianceModel>(pooling), max_approx != 0));
  c->tdp.reset(new TdpModel(*c->config, c->lexicon->get_silence_automaton().first_state()));
  return c;
}

void ref_destroy(void* h) { delete static_cast<RefCtx*>(h); }

uint32_t ref_num_states(void* h) { return static_cast<RefCtx*>(h)->lexicon->num_states(); }

// automaton of word w -> out (returns length)
uint32_t ref_word_automaton(void* h, uint32_t w, uint16_t* out) {
  MarkovAutomaton const& a = static_cast<RefCtx*>(h)->lexicon->get_automaton_for_word(w);
  for (size_t i = 0; i < a.num_states(); i++) out[i] = a[i];
  return a.num_states();
}

void ref_score_matrix(void* h, const float* feats, size_t T, double* out) {
  RefCtx* c = static_cast<RefCtx*>(h);
  std::vector<float> buf = padded(feats, T, c->dim);
  const size_t S = c->lexicon->num_states();
  FeatureIter it(buf.data(), c->dim);
  for (size_t t = 0; t < T; t++, ++it) {
    for (size_t s = 0; s < S; s++) out[t * S + s] = c->mixtures->score(it, s);
  }
}

// arg-min density per (t, s) from MixtureModel::min_score (Mixtures.cpp:696-713)
void ref_argmin_matrix(void* h, const float* feats, size_t T, uint16_t* out) {
  RefCtx* c = static_cast<RefCtx*>(h);
  std::vector<float> buf = padded(feats, T, c->dim);
  const size_t S = c->lexicon->num_states();
  FeatureIter it(buf.data(), c->dim);
  for (size_t t = 0; t < T; t++, ++it) {
    for (size_t s = 0; s < S; s++) out[t * S + s] = c->mixtures->min_score(it, s).second;
  }
}

// returns number of recognised words written to out_words (capacity T)
size_t ref_decode_pruned(void* h, const float* feats, size_t T, uint64_t* out_words) {
  RefCtx* c = static_cast<RefCtx*>(h);
  std::vector<float> buf = padded(feats, T, c->dim);
  Recognizer rec(*c->config, *c->lexicon, *c->mixtures, *c->tdp);
  std::vector<WordIdx> words;
  rec.recognizeSequence_pruned(FeatureIter(buf.data(), c->dim), FeatureIter(buf.data() + T * c->dim, c->dim), words);
  for (size_t i = 0; i < words.size(); i++) out_words[i] = words[i];
  return words.size();
}

void ref_edit_distance(void* h, const uint64_t* ref, size_t n_ref, const uint64_t* hyp, size_t n_hyp, uint16_t out4[4]) {
  RefCtx* c = static_cast<RefCtx*>(h);
  Recognizer rec(*c->config, *c->lexicon, *c->mixtures, *c->tdp);
  std::vector<WordIdx> r(ref, ref + n_ref), y(hyp, hyp + n_hyp);
  EDAccumulator ed = rec.editDistance(r.begin(), r.end(), y.begin(), y.end());
  out4[0] = ed.total_count; out4[1] = ed.substitute_count; out4[2] = ed.insert_count; out4[3] = ed.delete_count;
}

static double run_align(RefCtx* c, const float* feats, size_t T, const uint16_t* ref, size_t N, int pruned,
                        double thr, uint16_t* out_states) {
  std::vector<float> buf = padded(feats, T, c->dim);
  MarkovAutomaton automaton;
  automaton.states.assign(ref, ref + N);
  Alignment alignment(T);
  Aligner aligner(*c->mixtures, *c->tdp, 1);
  FeatureIter fb(buf.data(), c->dim), fe(buf.data() + T * c->dim, c->dim);
  AlignmentIter ab(&alignment[0], 1), ae(&alignment[0] + T, 1);
  double cost = pruned ? aligner.align_sequence_pruned(fb, fe, automaton, ab, ae, thr)
                       : aligner.align_sequence_full(fb, fe, automaton, ab, ae);
  for (size_t t = 0; t < T; t++) out_states[t] = alignment[t].state;
  return cost;
}

double ref_align_full(void* h, const float* feats, size_t T, const uint16_t* ref, size_t N, uint16_t* out_states) {
  return run_align(static_cast<RefCtx*>(h), feats, T, ref, N, 0, 0.0, out_states);
}

double ref_align_pruned(void* h, const float* feats, size_t T, const uint16_t* ref, size_t N, double thr,
                        uint16_t* out_states) {
  return run_align(static_cast<RefCtx*>(h), feats, T, ref, N, 1, thr, out_states);
}

// MixtureModel::accumulate (Mixtures.cpp:278-372) on a state path, then MixtureModel::write (:834-878) to `out_path`:
// the written MIXSET holds the accumulators (the reference keeps them private otherwise).
void ref_accumulate_and_write(void* h, const float* feats, size_t T, const uint16_t* states, int first_pass, int max_approx,
                              const char* out_path) {
  RefCtx* c = static_cast<RefCtx*>(h);
  std::vector<float> buf = padded(feats, T, c->dim);
  Alignment alignment(T);
  for (size_t t = 0; t < T; t++) alignment[t] = AlignmentItem(1, states[t], 1.0f);
  c->mixtures->accumulate(ConstAlignmentIter(&alignment[0], 1), ConstAlignmentIter(&alignment[0] + T, 1),
                          FeatureIter(buf.data(), c->dim), FeatureIter(buf.data() + T * c->dim, c->dim), first_pass != 0,
                          max_approx != 0);
  std::ofstream out(out_path, std::ios_base::out | std::ios_base::trunc | std::ios_base::binary);
  c->mixtures->write(out);
}

// ---- real-data helpers (oracle/gen_real_golden.py): the reference's own corpus reader, front-end
// post-processing (Corpus::read -> SignalAnalysis::process_features, Corpus.cpp:89-111) and GMM trainer
// (Trainer::train, Training.cpp:44-235) on SieTill features from /root/reference/data/new_features.

namespace {
struct RefCorpus {
  std::unique_ptr<Configuration> config;
  Lexicon lexicon;
  std::unique_ptr<SignalAnalysis> analyzer;
  std::unique_ptr<CorpusDescription> description;
  Corpus corpus;
};

RefCorpus* open_corpus(const char* config_path) {
  RefCorpus* c = new RefCorpus();
  c->config.reset(new Configuration(std::string(config_path)));
  c->lexicon = build_sietill_lexicon();
  c->description.reset(new CorpusDescription(*c->config));
  c->description->read(c->lexicon);
  c->analyzer.reset(new SignalAnalysis(*c->config));
  const ParameterString paramNormalizationPath("normalization-path", "");
  const ParameterString paramFeaturePath("feature-path", "");
  std::string norm = paramNormalizationPath(*c->config);
  if (norm.size() > 0) {
    std::ifstream in(norm.c_str(), std::ios_base::in);
    c->analyzer->read_normalization_file(in);  // SieTill.cpp:83-90
  }
  c->corpus.read(*c->description, paramFeaturePath(*c->config), *c->analyzer);
  return c;
}
}  // namespace

// SieTill.cpp:107-124 with action "train"; pooling as MixtureModel::VarianceModel
int ref_real_train(const char* config_path, int pooling, int max_approx) {
  RefCorpus* c = open_corpus(config_path);
  TdpModel tdp_model(*c->config, c->lexicon.get_silence_automaton()[0ul]);
  MixtureModel mixtures(*c->config, c->analyzer->n_features_total, c->lexicon.num_states(),
                        static_cast<MixtureModel::VarianceModel>(pooling), max_approx != 0);
  Trainer trainer(*c->config, c->lexicon, mixtures, tdp_model, max_approx != 0);
  trainer.train(c->corpus);
  delete c;
  return 0;
}

void* ref_corpus_open(const char* config_path) { return open_corpus(config_path); }
void ref_corpus_close(void* h) { delete static_cast<RefCorpus*>(h); }
size_t ref_corpus_size(void* h) { return static_cast<RefCorpus*>(h)->corpus.get_corpus_size(); }
size_t ref_corpus_dim(void* h) { return static_cast<RefCorpus*>(h)->corpus.get_features_per_timeframe(); }
size_t ref_corpus_frames(void* h, size_t s) {
  std::pair<FeatureIter, FeatureIter> f = static_cast<RefCorpus*>(h)->corpus.get_feature_sequence(s);
  return f.second - f.first;
}
// copies the processed features of segment s and its reference word sequence; returns the word count
size_t ref_corpus_get(void* h, size_t s, float* feats, uint64_t* words) {
  RefCorpus* c = static_cast<RefCorpus*>(h);
  std::pair<FeatureIter, FeatureIter> f = c->corpus.get_feature_sequence(s);
  std::memcpy(feats, *f.first, sizeof(float) * (f.second - f.first) * c->corpus.get_features_per_timeframe());
  std::pair<WordIter, WordIter> w = c->corpus.get_word_sequence(s);
  size_t n = 0;
  for (WordIter it = w.first; it != w.second; ++it) words[n++] = *it;
  return n;
}

}  // extern "C"
