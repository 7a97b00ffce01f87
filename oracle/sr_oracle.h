/*
 * sr_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's GMM-scoring + Viterbi hot path
 * (kkromberg/SpeechRecognition, src/sietill/{Mixtures,Recognizer,Alignment,
 * TdpModel,Lexicon}.cpp).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (libsrgpu.so) never links or calls it.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the
 * reference itself, compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/libsietill_ref.so (tests/test_oracle_vs_reference.py), and
 * against the golden vectors that build produced (tests/golden/ npz files,
 * generator: oracle/gen_golden.py).
 *
 * All scores are negative natural-log likelihoods (costs), IEEE double.
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_POOL_GLOBAL = 0, ORC_POOL_MIXTURE = 1, ORC_POOL_NONE = 2 }; /* Mixtures.hpp:20-24 */

typedef struct orc_model orc_model;

/* MIXSET v2 loader + finalize (Mixtures.cpp:748-830, 374-461, 251-275). NULL on error. */
orc_model* orc_model_load(const char* path, uint32_t dim, int pooling, int max_approx);
void orc_model_free(orc_model* m);
const char* orc_last_error(void);

uint32_t orc_model_dim(const orc_model* m);
uint32_t orc_model_num_states(const orc_model* m);     /* mixtures */
uint32_t orc_model_num_means(const orc_model* m);
uint32_t orc_model_num_vars(const orc_model* m);
uint32_t orc_model_num_densities(const orc_model* m);  /* sum over mixtures */
/* derived tables (finalize): means[n_mean*D], vars_inv[n_var*D], norm[n_var], logw[n_mean] */
const double* orc_model_means(const orc_model* m);
const double* orc_model_vars_inv(const orc_model* m);
const double* orc_model_norm(const orc_model* m);
const double* orc_model_logw(const orc_model* m);
/* flattened mixtures: offsets[S+1]; per density mean_idx / var_idx (uint32) */
const uint32_t* orc_model_mix_offsets(const orc_model* m);
const uint32_t* orc_model_mix_mean_idx(const orc_model* m);
const uint32_t* orc_model_mix_var_idx(const orc_model* m);

/* MixtureModel::score (Mixtures.cpp:737-744): min_score (:696-713) or sum_score (:719-728). */
double orc_score(const orc_model* m, const float* x, uint32_t state);
/* same + arg-min density within the mixture (min_score's .second); sum mode returns 0 */
double orc_score_argmin(const orc_model* m, const float* x, uint32_t state, uint32_t* density);
/* dense [T x S] table, row-major; n_threads<=1 -> serial, else OpenMP over frames */
void orc_score_matrix(const orc_model* m, const float* feats, size_t T, double* out, int n_threads);

/* Flattened lexicon (Lexicon.cpp:11-22, MarkovAutomaton.hpp:22-28):
 * word w owns automaton[word_off[w] .. word_off[w+1]) (uint16 state ids, repetitions expanded). */
typedef struct {
  uint32_t n_words;
  uint32_t n_states;        /* Lexicon::num_states() */
  uint32_t silence_idx;     /* word index of silence */
  const uint32_t* word_off; /* [n_words+1] */
  const uint16_t* automaton;
} orc_lexicon;

typedef struct {
  double loop, forward, skip; /* TdpModel.cpp:5-7 */
  uint16_t silence_state;     /* TdpModel.hpp:19 */
} orc_tdp;

/* TdpModel::score (TdpModel.cpp:19-29) */
double orc_tdp_score(const orc_tdp* tdp, uint16_t to, size_t jump);

typedef struct {
  double am_threshold;  /* Recognizer.cpp:31 */
  double word_penalty;  /* Recognizer.cpp:32 */
} orc_search_params;

/* Recognizer::recognizeSequence_pruned (Recognizer.cpp:103-232).
 * Scores come lazily from `m` (am_cache like the reference) unless `dense` != NULL, in which case
 * am(t, s) = dense[t*dense_stride + s] (used to test a decoder in isolation).
 * out_words: capacity >= T; returns number of words written.
 * tb_score/tb_word/tb_bkp: optional [T+1] dumps of the traceback array (may be NULL).
 * n_scored (optional): number of scorer_.score() calls made (lazy-scoring statistic). */
size_t orc_decode_pruned(const orc_model* m, const double* dense, size_t dense_stride,
                         const orc_lexicon* lex, const orc_tdp* tdp, const orc_search_params* sp,
                         const float* feats, size_t T, uint32_t dim,
                         uint32_t* out_words,
                         double* tb_score, uint16_t* tb_word, uint16_t* tb_bkp,
                         uint64_t* n_scored);

/* Aligner::align_sequence_full (Alignment.cpp:50-144). reference automaton `ref` (N uint16 ids).
 * out_states[T]. Returns path cost. Requires 1 <= N <= T (the reference overflows its cost
 * arrays otherwise) -- returns NaN and sets orc_last_error() if violated. */
double orc_align_full(const orc_model* m, const double* dense, size_t dense_stride,
                      const orc_tdp* tdp, const uint16_t* ref, size_t N,
                      const float* feats, size_t T, uint32_t dim, uint16_t* out_states);

/* Aligner::align_sequence_pruned (Alignment.cpp:149-288). */
double orc_align_pruned(const orc_model* m, const double* dense, size_t dense_stride,
                        const orc_tdp* tdp, const uint16_t* ref, size_t N,
                        const float* feats, size_t T, uint32_t dim, double pruning_threshold,
                        uint16_t* out_states);

/* Recognizer::editDistance (Recognizer.cpp:332-389), bug-compatible (16-bit counters, the
 * row-0 insertion counter quirk). out4 = {total, substitutions, insertions, deletions}. */
void orc_edit_distance(const uint64_t* ref, size_t n_ref, const uint64_t* hyp, size_t n_hyp, uint16_t out4[4]);

/* MixtureModel::accumulate (Mixtures.cpp:278-372) after reset_accumulators (:235-247): EM statistics of a
 * state path (`states[t]` = aligned mixture of frame t, one item per frame).  first_pass: every frame goes to
 * density 0 of its mixture; else max_approx: to the arg-min density (min_score); else soft memberships
 * exp(-score)/sum with entries below 1e-8 skipped.  Outputs are indexed like the model's mean / variance
 * tables: mean_acc[n_mean*D], mean_w[n_mean], var_acc[n_var*D] (starts at 1e-4, :243), var_w[n_var]. */
void orc_accumulate(const orc_model* m, const float* feats, size_t T, const uint16_t* states, int first_pass,
                    int max_approx, double* mean_acc, double* mean_w, double* var_acc, double* var_w);

/* Corpus-level loop like Recognizer::recognize (Recognizer.cpp:38-92): decodes utterances
 * [0,n_utts) with `n_threads` OpenMP threads (schedule(dynamic), like :46); returns wall seconds
 * of the utterance loop (the reference's timed region, :45-80). frame_off[n_utts+1] in frames.
 * out_words/out_word_off as in the GPU API (out_words capacity = total frames). */
double orc_recognize_batch(const orc_model* m, const orc_lexicon* lex, const orc_tdp* tdp,
                           const orc_search_params* sp, const float* feats, const uint64_t* frame_off,
                           size_t n_utts, uint32_t dim, int n_threads,
                           uint32_t* out_words, uint64_t* out_word_off);

/* ---- B1: bigram linear-lexicon beam search (rwth-asr-0.5/src/Teaching/LinearSearch.cc) ------------------------
 * PARITY UNPINNED: the RWTH toolkit cannot be built here and holds no fixtures for this decoder, so this
 * restatement -- written line by line after LinearSearch.cc:211-436,496-515 and BookKeeping.cc -- is the only
 * specification the GPU kernel is tested against.
 * Scores are float (Teaching/Types.hh:17).  words 0..W-1 (one of them `silence`); slot w+W is the silence copy entered
 * after word w.  word_off[W+1]/mixtures[]: the linear lexicon (mixture = emission state per position);
 * lm[w*W + h] = -log p(w | h); tdp[isSilence][0..2] loop/forward/skip, tdp[isSilence][3] exit penalty
 * (SearchSpace::setTransitionScores :169-180).  am scores: (float)dense[t*stride + mixture].
 * Pruning thresholds >= FLT_MAX switch the beam off (:445-455,499-510).
 * Output: the traceback items (word, score, time) of LinearSearch::getResult; returns their number (written up to
 * `cap`).  stats (optional, [4]): sum over frames of word ends after merging, active words, state hyps, book entries. */
size_t orc_bigram_decode(const double* dense, size_t dense_stride, size_t T, uint32_t W, uint32_t silence,
                         const uint32_t* word_off, const uint16_t* mixtures, const float* lm, const float tdp[2][4],
                         float acoustic_pruning, float lm_pruning, uint32_t* out_word, float* out_score,
                         uint32_t* out_time, size_t cap, uint64_t* stats);

#ifdef __cplusplus
}
#endif
#endif
