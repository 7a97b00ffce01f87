// GpuTrainer.hpp -- reference-side binding for the TRAINING callers of the hot path: the three corpus-wide loops of
// Trainer::train (src/sietill/Training.cpp) that run the scorer and the aligner over every segment, each as ONE device pass.
// Written against the REFERENCE's headers (Alignment.hpp, Corpus.hpp, Lexicon.hpp, Mixtures.hpp, TdpModel.hpp, Types.hpp);
// it only compiles inside that tree.  tests/test_integration_stub.py compiles AND links it against the reference's own
// objects and libsrgpu.so where /root/reference is present.  Nothing in this repo's product path includes it.
//
//   GpuTrainer::realign         the re-alignment loop, Training.cpp:163-184 (Aligner::align_sequence_pruned / _full per segment,
//                               Alignment.hpp:49-55)                                   -> sr_align_corpus[_pruned]
//   GpuTrainer::calc_am_score   Trainer::calc_am_score, Training.cpp:585-612           -> sr_path_scores_corpus
//   GpuTrainer::accumulate      MixtureModel::accumulate, Mixtures.cpp:278-372         -> sr_accumulate_corpus
//   GpuTrainer::write / finalize  MixtureModel::write (:834-878) / finalize (:374-461) -> sr_mixset_write / sr_model_create_from_statistics
//
// The alignment buffers are the reference's own: `Alignment` = std::vector<AlignmentItem> with num_max_aligns items per frame
// (Types.hpp:29-40, Training.cpp:120-123); the device returns one state per frame, stored as item 0 {count 1, state, weight 1}
// exactly like Aligner::align_sequence_* leaves it (Alignment.cpp:131-134).
#ifndef __GPU_TRAINER_HPP__
#define __GPU_TRAINER_HPP__

#include <stdexcept>
#include <string>
#include <vector>

#include "Alignment.hpp"
#include "Corpus.hpp"
#include "GpuMixtureScorer.hpp"
#include "Lexicon.hpp"
#include "MarkovAutomaton.hpp"
#include "TdpModel.hpp"
#include "Types.hpp"
#include "srgpu.h"

class GpuTrainer {
public:
  // tdp_loop/forward/skip: TdpModel keeps them private (TdpModel.hpp:25-28), so they are passed again ("tdp-loop" ... in the
  // config, TdpModel.cpp:5-7); num_max_aligns, pruning_threshold, alignment_pruning as Trainer reads them (Training.cpp:24-39)
  GpuTrainer(GpuMixtureScorer& scorer, Lexicon const& lexicon, TdpModel const& tdp_model,
             double tdp_loop, double tdp_forward, double tdp_skip,
             size_t num_max_aligns, double pruning_threshold, bool alignment_pruning)
            : scorer_(scorer), lexicon_(lexicon), silence_state_(tdp_model.silence_state),
              num_max_aligns_(num_max_aligns), pruning_threshold_(pruning_threshold), alignment_pruning_(alignment_pruning),
              corpus_(NULL), resident_(NULL) {
    tdp_[0] = tdp_loop; tdp_[1] = tdp_forward; tdp_[2] = tdp_skip;
  }
  ~GpuTrainer() { sr_corpus_destroy(corpus_); }

  // Trainer::build_segment_automaton (Training.cpp:238-253)
  MarkovAutomaton build_segment_automaton(WordIter segment_begin, WordIter segment_end) const {
    std::vector<MarkovAutomaton const*> automata;
    for (WordIter iter = segment_begin; iter != segment_end; iter++) {
      automata.push_back(&lexicon_.get_silence_automaton());
      automata.push_back(&lexicon_.get_automaton_for_word(*iter));
    }
    automata.push_back(&lexicon_.get_silence_automaton());
    return MarkovAutomaton().concat(automata);
  }

  // the re-alignment loop (Training.cpp:163-184) for every segment at once; `alignment` as Trainer::train allocates it
  // (total frames x num_max_aligns items); returns the per-segment path costs
  std::vector<double> realign(Corpus const& corpus, std::vector<MarkovAutomaton> const& segment_automata, Alignment& alignment) {
    upload(corpus);
    const size_t n = corpus.get_corpus_size();
    std::vector<uint16_t> automata;
    std::vector<uint64_t> aut_off(1, 0u);
    for (SegmentIdx s = 0ul; s < n; s++) {
      MarkovAutomaton const& a = segment_automata[s];
      for (size_t i = 0ul; i < a.num_states(); i++) automata.push_back(a[i]);
      aut_off.push_back(automata.size());
    }
    const size_t frames = corpus.get_total_frame_count() / corpus.get_features_per_timeframe();
    std::vector<uint16_t> states(frames + 1u);
    std::vector<double> cost(n + 1u);
    const int rc = alignment_pruning_
        ? sr_align_corpus_pruned(scorer_.handle(), corpus_, automata.data(), aut_off.data(), tdp_, silence_state_,
                                 pruning_threshold_, SR_GMM_DEFAULT, states.data(), cost.data())
        : sr_align_corpus(scorer_.handle(), corpus_, automata.data(), aut_off.data(), tdp_, silence_state_,
                          SR_GMM_DEFAULT, states.data(), cost.data());
    check(rc);
    alignment.resize(frames * num_max_aligns_);
    for (size_t t = 0ul; t < frames; t++) {
      alignment[t * num_max_aligns_] = AlignmentItem(1u, states[t], 1.0f);
    }
    cost.resize(n);
    return cost;
  }

  // Trainer::calc_am_score (Training.cpp:585-612): the sum stays sequential on the host, in the reference's order
  double calc_am_score(Corpus const& corpus, Alignment const& alignment) {
    upload(corpus);
    const size_t frames = corpus.get_total_frame_count() / corpus.get_features_per_timeframe();
    std::vector<uint16_t> states(frames + 1u);
    for (size_t t = 0ul; t < frames; t++) states[t] = alignment[t * num_max_aligns_].state;
    std::vector<double> per_frame(frames + 1u);
    check(sr_path_scores_corpus(scorer_.handle(), corpus_, states.data(), SR_GMM_DEFAULT, per_frame.data()));
    double total_score = 0.0;
    for (size_t t = 0ul; t < frames; t++) total_score += per_frame[t];
    return total_score / frames;
  }

  // MixtureModel::accumulate over the whole corpus (Training.cpp:200 calls it with all features and the whole alignment);
  // the statistics stay in this object until write() / finalize()
  void accumulate(Corpus const& corpus, Alignment const& alignment, bool first_pass, bool max_approx) {
    upload(corpus);
    const size_t frames = corpus.get_total_frame_count() / corpus.get_features_per_timeframe();
    std::vector<uint16_t> states(frames + 1u);
    for (size_t t = 0ul; t < frames; t++) states[t] = alignment[t * num_max_aligns_].state;
    uint32_t dim, n_states, n_mean, n_var; uint64_t n_dens;
    check(sr_model_info(scorer_.handle(), &dim, &n_states, &n_dens));
    check(sr_model_tying_info(scorer_.handle(), &n_mean, &n_var));
    mean_acc_.assign((size_t)n_mean * dim, 0.0); mean_w_.assign(n_mean, 0.0);
    var_acc_.assign((size_t)n_var * dim, 0.0);   var_w_.assign(n_var, 0.0);
    check(sr_accumulate_corpus(scorer_.handle(), corpus_, states.data(), first_pass, max_approx,
                               mean_acc_.data(), mean_w_.data(), var_acc_.data(), var_w_.data()));
    dens_off_.assign(n_states + 1u, 0u); dens_mean_.assign(n_dens, 0u); dens_var_.assign(n_dens, 0u);
    check(sr_model_topology(scorer_.handle(), dens_off_.data(), dens_mean_.data(), dens_var_.data()));
    dim_ = dim;
  }

  // MixtureModel::write (Mixtures.cpp:834-878) of the accumulated statistics: byte for byte the file the reference writes
  // after the same accumulate(), so that MixtureModel::read (:748-830) -- or the next GpuMixtureScorer -- can load it
  void write(std::string const& path) const {
    check(sr_mixset_write(path.c_str(), dim_, dens_off_.size() - 1u, dens_off_.data(), mean_w_.size(), var_w_.size(),
                          dens_mean_.data(), dens_var_.data(), mean_acc_.data(), mean_w_.data(), var_acc_.data(), var_w_.data()));
  }

  // MixtureModel::finalize (Mixtures.cpp:374-461) of the accumulated statistics -> a new device model (the caller owns it)
  sr_model* finalize(int pooling, bool max_approx, int device = 0) const {
    sr_model* m = NULL;
    check(sr_model_create_from_statistics(device, dim_, dens_off_.size() - 1u, dens_off_.data(), mean_w_.size(), var_w_.size(),
                                          dens_mean_.data(), dens_var_.data(), mean_acc_.data(), mean_w_.data(), var_acc_.data(),
                                          var_w_.data(), pooling, max_approx, &m));
    return m;
  }

private:
  static void check(int rc) {
    if (rc != SR_OK) throw std::runtime_error(sr_last_error());
  }

  // the corpus stays resident on the device across the passes of one training iteration (Training.cpp:158-213 touches the same
  // features in re-alignment, accumulate and calc_am_score)
  void upload(Corpus const& corpus) {
    if (resident_ == &corpus) return;
    sr_corpus_destroy(corpus_);
    corpus_ = NULL;
    const size_t n = corpus.get_corpus_size(), dim = corpus.get_features_per_timeframe();
    std::vector<uint64_t> frame_off(n + 1u, 0u);
    for (SegmentIdx s = 0ul; s < n; s++) frame_off[s + 1u] = corpus.get_feature_offsets(s).second / dim;  // FLOAT offsets, Corpus.cpp:104
    check(sr_corpus_upload(scorer_.handle(), *corpus.get_all_features().first, frame_off.data(), n, &corpus_));
    resident_ = &corpus;
  }

  GpuMixtureScorer& scorer_;
  Lexicon const&    lexicon_;
  StateIdx          silence_state_;
  size_t            num_max_aligns_;
  double            pruning_threshold_;
  bool              alignment_pruning_;
  double            tdp_[3];
  sr_corpus*        corpus_;
  Corpus const*     resident_;
  uint32_t          dim_;
  std::vector<uint32_t> dens_off_, dens_mean_, dens_var_;
  std::vector<double>   mean_acc_, mean_w_, var_acc_, var_w_;
};

#endif /* __GPU_TRAINER_HPP__ */
