/*
 * srgpu.h -- C ABI of libsrgpu.so: MI355X (gfx950) GMM acoustic scorer + Viterbi decoder/aligner.
 *
 * Drop-in boundary for the hot path of kkromberg/SpeechRecognition's `sietill` recogniser.
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository root, src/sietill/...).  Plain pointers and sizes only; no C++/torch types; all
 * functions return 0 on success or a negative SR_E* code, with a thread-local message in
 * sr_last_error().  No exceptions cross this boundary.
 *
 * Conventions (same as the reference):
 *   scores   negative natural-log likelihoods (costs, lower is better), IEEE double
 *   features float32, row-major [frames x dim], utterances concatenated (Corpus layout,
 *            Corpus.cpp:89-111); offsets here are in FRAMES, not floats
 *   states   mixture (= HMM state) indices; words are indices into the lexicon
 *
 * Ownership: the caller owns every host buffer; handles own their device memory.  Model and
 * lexicon are immutable after creation.  A handle is bound to one HIP device and must be used by
 * one host thread at a time (one handle per GPU; utterance batches shard across GPUs with no
 * collective, Recognizer.cpp:46-47).
 */
#ifndef SRGPU_H
#define SRGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SR_API __attribute__((visibility("default")))
#else
#define SR_API
#endif

/* ABI version of this header.  Bumped whenever a struct below grows or changes layout or an entry point changes its signature
 * (4: sr_bigram_params gained `flags` in round 3 -- a caller compiled against the 16-byte struct of version 3 would be read 4 bytes
 * past its end; sr_model_trim, SR_GMM_DEFAULT).  A binding checks `sr_abi_version() == SR_ABI_VERSION` once after loading the
 * library (capi.py and sr_sietill.hpp do) and zero-initialises the parameter structs field by field. */
#define SR_ABI_VERSION 4
SR_API int sr_abi_version(void);

#define SR_OK 0
#define SR_EINVAL (-1)   /* bad argument (message says which) */
#define SR_EHIP (-2)     /* HIP runtime error */
#define SR_ENODEV (-3)   /* no usable gfx950 device */
#define SR_ELIMIT (-4)   /* size outside what the kernels support (message says which) */
#define SR_ENOMEM (-5)   /* host allocation failed (std::bad_alloc caught at the boundary) */
#define SR_EINTERNAL (-6) /* any other C++ exception caught at the boundary (message carries what()) */
#define SR_ECORRUPT (-7) /* a traceback that does not walk back to frame 0 (a back pointer that does not fall, a word outside the
                           lexicon): no words are reported for the call */

/* GMM scoring kernels (MixtureModel::score, Mixtures.cpp:737-744) */
#define SR_GMM_MFMA 0   /* dense FP64 MFMA contraction + fused min / -log-sum-exp epilogue (<= 1e-9 relative, ~1e-15 on
                           well-conditioned models; what SR_GMM_DEFAULT picks for sum scoring).  Dimension <= 63; a model of
                           dimension 64 .. 160 (the largest the library takes) is scored by SR_GMM_EXACT's kernel instead */
#define SR_GMM_EXACT 1  /* direct form replaying density_score_sse's operation order (Mixtures.cpp:645-690): bit-exact */
#define SR_GMM_PREFILTER 2  /* bit-exact like SR_GMM_EXACT: a 16-bit MFMA prefilter selects the densities that can be the
                               minimum, FP64 replays only those.  Max-approx models with <= 128 densities per mixture and
                               dim <= 62; any other model is scored by SR_GMM_EXACT's kernel instead (same bits). */

#define SR_GMM_DEFAULT 3  /* what a drop-in caller wants: the fastest kernel that reproduces MixtureModel::score for THIS model --
                             SR_GMM_PREFILTER (bit-exact) for max-approx models; for sum scoring (max-approx false,
                             Mixtures.cpp:719-728) the FP64-MFMA kernel with its fused -log sum exp, i.e. SR_GMM_MFMA's bound:
                             <= 1e-9 relative (measured 6e-16 on synthetic models, 2.3e-9 worst on a trained real-speech model;
                             the direct form SR_GMM_EXACT is 1e-12 from the reference's libm in sum mode and 3x slower -- a
                             caller that wants those three digits passes SR_GMM_EXACT).  Words and tracebacks through this
                             selector are held to the reference on the sum-mode goldens (tests/test_gpu_parity.py).
                             The aligner / path-score entry points score only the states they need with the bit-exact kernel. */

typedef struct sr_model sr_model;     /* replaces MixtureModel as a FeatureScorer (Mixtures.hpp:18, FeatureScorer.hpp:12-16) */
typedef struct sr_corpus sr_corpus;   /* replaces Corpus' feature store (Corpus.hpp:79-84, Corpus.cpp:141-144) */
typedef struct sr_lexicon sr_lexicon; /* replaces Lexicon + TdpModel as search network (Lexicon.hpp:16-33, TdpModel.hpp:13-29) */

SR_API const char* sr_last_error(void);
SR_API int sr_device_count(int* count);

/* ---- model ------------------------------------------------------------------------------------
 * Finalised tables as MixtureModel holds them after read()+finalize() (Mixtures.cpp:374-461,
 * Mixtures.hpp:71-84), expanded per density (tied variances repeated):
 *   dens_off[n_states+1]  densities of state s are rows dens_off[s] .. dens_off[s+1]-1
 *   means, inv_vars       [C x dim] doubles (means_, vars_inv_);  norm, logw  [C] (norm_, mean_weights_log_)
 *   max_approx            1: min_score (Mixtures.cpp:696-713), 0: sum_score (:719-728)
 * Limits: dim <= 160 (the matrix-core kernels end at 63 and 62: the exact kernel scores what lies beyond), C < 2^31.  (The reference
 * itself stops at 65535 densities, Mixtures.cpp:766.) */
SR_API int sr_model_create(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off,
                    const double* means, const double* inv_vars, const double* norm, const double* logw,
                    int max_approx, sr_model** out);
/* MixtureModel(config, dim, ...) with "load-mixtures-from" (Mixtures.cpp:156-174): reads a MIXSET v2
 * file (read(), :748-830), derives the tables (finalize(), :374-461) on the host and creates the
 * device model.  pooling: 0 global, 1 mixture, 2 none (MixtureModel::VarianceModel, Mixtures.hpp:20-24).
 * Malformed files return SR_EINVAL with the reference's message instead of abort() (Mixtures.cpp:97-102). */
SR_API int sr_model_load_mixset(const char* path, uint32_t dim, int pooling, int max_approx, int device, sr_model** out);
/* MixtureModel::finalize (Mixtures.cpp:374-461) on in-memory statistics -- the model update of an EM iteration:
 * accumulators as sr_accumulate_corpus returns them (+ the topology: dens_off, accumulator rows per density) ->
 * new device model.  The divisions, variances and per-density tables are computed on the device, in the reference's
 * operation order; the logarithms (norm_, mean_weights_log_) go through the host's libm so that the tables carry the
 * reference's bits (the variances come back to the host for that).  sr_mixset_write stores the same statistics as a MIXSET v2 file exactly like
 * MixtureModel::write (Mixtures.cpp:834-878: unreferenced rows dropped and renumbered). */
SR_API int sr_model_create_from_statistics(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off,
                                           uint32_t n_mean, uint32_t n_var, const uint32_t* dens_mean, const uint32_t* dens_var,
                                           const double* mean_acc, const double* mean_w, const double* var_acc, const double* var_w,
                                           int pooling, int max_approx, sr_model** out);
SR_API int sr_mixset_write(const char* path, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean,
                           uint32_t n_var, const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc,
                           const double* mean_w, const double* var_acc, const double* var_w);
SR_API int sr_model_destroy(sr_model* m);
SR_API int sr_model_info(const sr_model* m, uint32_t* dim, uint32_t* n_states, uint64_t* n_densities);

/* ---- corpus / frame-batch feeder (Corpus.cpp:89-111) --------------------------------------------
 * Copies n_utts concatenated utterances to the device. frame_off[n_utts+1], frame_off[0] == 0.
 * The host buffer is borrowed for the duration of the call only. */
SR_API int sr_corpus_upload(sr_model* m, const float* feats, const uint64_t* frame_off, uint32_t n_utts, sr_corpus** out);
/* The frame-batch feeder proper: like sr_corpus_upload, but it returns as soon as the handle exists.  A feeder thread copies
 * the host buffer in 2 MiB pieces through two pinned staging buffers on the corpus' own copy stream; the compute entry
 * points below wait -- on the device, per score chunk -- only for the pieces a chunk needs, so the transfer of later
 * utterances overlaps the scoring of earlier ones.  The host buffer is borrowed until sr_corpus_wait() has returned (or
 * the corpus has been destroyed); sr_corpus_wait returns the feeder's status.  sr_recognize_batch feeds this way. */
SR_API int sr_corpus_upload_async(sr_model* m, const float* feats, const uint64_t* frame_off, uint32_t n_utts, sr_corpus** out);
SR_API int sr_corpus_wait(sr_corpus* c);
/* A corpus must be destroyed BEFORE the model it was uploaded to.  Its search-path device buffers (features, frame offsets, word and
 * traceback outputs) are not freed but parked on the model for the next corpus -- sr_recognize_batch creates and destroys one per
 * call, and nine allocations per batch cost about a millisecond -- if they hold at most SRGPU_SPARE_MB MiB together (default 256;
 * 0 keeps nothing); one set is kept, until the next upload adopts it, sr_model_trim() or sr_model_destroy(). */
SR_API int sr_corpus_destroy(sr_corpus* c);
/* Releases what the model keeps for reuse between calls: the parked buffers above, and the scoring path's deferred-leftover segments
 * (models of more than 32 densities per mixture; up to SRGPU_DEFER_MB, 23 GB at 8000 states x 64 densities / 302 685 frames) --
 * the next scoring call allocates them again. */
SR_API int sr_model_trim(sr_model* m);

/* ---- scoring: FeatureScorer::prepare_sequence + score (FeatureScorer.hpp:14-15) -------------------
 * Dense table out[total_frames x n_states] (row-major, host memory): out[t*n_states+s] is what
 * MixtureModel::score(frame t, s) returns.  This is the table a `GpuMixtureScorer::prepare_sequence`
 * fills, exactly like NeuralNetwork::prepare_sequence does (NeuralNetwork.cpp:184-199). */
SR_API int sr_score_corpus(sr_model* m, sr_corpus* c, int gmm_kernel, double* out);
/* one-shot convenience over host features of a single sequence */
SR_API int sr_score_frames(sr_model* m, const float* feats, uint64_t n_frames, int gmm_kernel, double* out);

/* ---- search network ---------------------------------------------------------------------------
 * Linear whole-word lexicon as Lexicon::add_word builds it (Lexicon.cpp:11-22): word w owns
 * automaton[word_off[w] .. word_off[w+1]) (state ids with repetitions expanded,
 * MarkovAutomaton.hpp:22-28).  tdp = {loop, forward, skip} (TdpModel.cpp:5-7); silence_state as
 * TdpModel::silence_state.  The initial hypothesis sits at position 0 of word 0 like the
 * reference's (Recognizer.cpp:120), which is a word end when word 0 (normally silence) has one
 * position.  Limits: <= 65535 words, <= 65534 positions in total (beyond 8192 type-padded positions the search keeps its
 * hypotheses in device memory instead of LDS: same results, slower), some word with >= 2 positions. */
SR_API int sr_lexicon_create(sr_model* m, uint32_t n_words, const uint32_t* word_off, const uint16_t* automaton,
                      uint32_t silence_idx, const double tdp[3], uint16_t silence_state, sr_lexicon** out);
SR_API int sr_lexicon_destroy(sr_lexicon* l);
/* Which search kernel sr_recognize_corpus runs on this lexicon, as text (for logs and benchmark reports): "words <nw> x <lanes>
 * plain <L> [general]" = the word-per-lane kernel (viterbi_words.hip), "slots" = the slot-per-lane kernel (viterbi_fast.hip),
 * "big" = hypotheses in device memory.  out is NUL-terminated within cap bytes. */
SR_API int sr_lexicon_describe(const sr_lexicon* l, char* out, size_t cap);

typedef struct {
  double am_threshold; /* "am-threshold", Recognizer.cpp:31 (beam) */
  double word_penalty; /* "word-penalty", Recognizer.cpp:32 */
  int gmm_kernel;      /* SR_GMM_DEFAULT, or SR_GMM_MFMA / SR_GMM_EXACT / SR_GMM_PREFILTER */
  int flags;           /* 0, SR_SEARCH_GENERAL_KERNEL or SR_SEARCH_SLOT_KERNEL (was `reserved`, must be 0 otherwise) */
} sr_search_params;
/* Decode every utterance with the general kernel (slots in reference order, sequential boundary replay inline,
 * viterbi_decode.hip) instead of the type-sorted fast kernel: same results, several times slower.  The fast kernel hands
 * an utterance to it by itself when it meets a negative emission cost; this flag is for cross-checking the two. */
#define SR_SEARCH_GENERAL_KERNEL 1
/* Lexica whose words all have at most four positions (and at most 3072 words, score rows that fit the LDS twice) are searched
 * by the word-per-lane kernel (viterbi_words.hip: a word's hypotheses live in registers) -- also where they have more than the
 * 8192 type-padded positions the slot-per-lane kernel holds (2 731 .. 3 072 three-state words; round 4); this flag keeps them on the
 * slot-per-lane kernel that serves every other lexicon (viterbi_fast.hip).  Same results; for cross-checking the two. */
#define SR_SEARCH_SLOT_KERNEL 2

/* ---- decoder: Recognizer::recognize / recognizeSequence_pruned (Recognizer.cpp:38-92, :103-232) ---
 * Scores every frame of the resident corpus and runs the beam Viterbi per utterance, all on the
 * device.  out_words (capacity: total frames, a safe upper bound) receives the recognised word
 * indices of all utterances back to back, silence removed; utterance u owns
 * out_words[out_word_off[u] .. out_word_off[u+1]).
 * Optional traceback dump (may be NULL), each [total_frames + n_utts]: entry frame_off[u]+u+t is
 * traceback[t] of utterance u, t = 0..T_u (Recognizer.cpp:118,191-208). */
SR_API int sr_recognize_corpus(sr_model* m, sr_corpus* c, sr_lexicon* l, const sr_search_params* p,
                        uint32_t* out_words, uint64_t* out_word_off,
                        double* tb_score, uint16_t* tb_word, uint16_t* tb_bkp);
/* The traceback loop alone (Recognizer.cpp:222-231: `t = T; while (t > 0) { push word unless silence; t = traceback[t].bkp }`,
 * reversed), with the checks every search kernel here applies before it follows an entry: bkp < t (the reference writes
 * bkp = t - 1 of an earlier frame, :140,170, so t falls strictly), word < n_words, at most T words.  SR_ECORRUPT otherwise.
 * sr_traceback_corpus walks caller-supplied dumps in sr_recognize_corpus' layout ([total_frames + n_utts], words not slots)
 * on the device, with the kernels' own walker; sr_traceback_words walks one utterance's T + 1 entries on the host. */
SR_API int sr_traceback_corpus(sr_model* m, sr_corpus* c, sr_lexicon* l, const uint16_t* tb_word, const uint16_t* tb_bkp,
                               uint32_t* out_words, uint64_t* out_word_off);
SR_API int sr_traceback_words(uint32_t n_frames, const uint16_t* tb_word, const uint16_t* tb_bkp, uint32_t silence_word,
                              uint32_t n_words, uint32_t* out_words, uint32_t* out_count);
/* one-shot convenience: upload + recognise + free */
SR_API int sr_recognize_batch(sr_model* m, sr_lexicon* l, const sr_search_params* p, const float* feats,
                       const uint64_t* frame_off, uint32_t n_utts, uint32_t* out_words, uint64_t* out_word_off);

/* ---- several devices: the `#pragma omp parallel for` over segments of Recognizer::recognize (Recognizer.cpp:46-47) -------
 * Utterances are independent, so a batch shards across devices with no collective: sr_shard_utterances deals them by
 * greedy longest-processing-time on frame counts (decreasing length, each to the lightest shard so far; deterministic);
 * shard_of_utt[n_utts] receives the shard of every utterance, shard_frames[n_shards] (may be NULL) the frames per shard.
 * sr_recognize_batch_multi runs one host thread per (model, lexicon) replica -- models[d] on its own device, or several
 * replicas on one device -- which feeds (sr_corpus_upload_async, straight from the caller's buffer) and recognises its
 * shard; the word sequences come back in corpus order, exactly what sr_recognize_batch returns on one device. */
SR_API int sr_shard_utterances(const uint64_t* frame_off, uint32_t n_utts, uint32_t n_shards, uint32_t* shard_of_utt,
                               uint64_t* shard_frames);
SR_API int sr_recognize_batch_multi(sr_model* const* models, sr_lexicon* const* lexica, uint32_t n_devices,
                                    const sr_search_params* p, const float* feats, const uint64_t* frame_off, uint32_t n_utts,
                                    uint32_t* out_words, uint64_t* out_word_off, uint64_t* shard_frames);

/* ---- forced aligner: Aligner::align_sequence_full (Alignment.cpp:50-144) --------------------------
 * Utterance u is aligned against automata[aut_off[u] .. aut_off[u+1]) (N_u state ids; training
 * builds `sil w1 sil ... sil`, Training.cpp:239-253).  Requires 1 <= N_u <= T_u (the reference
 * indexes T-sized cost arrays by position).  out_states[total_frames] receives
 * AlignmentItem::state per frame (count = 1, weight = 1 implied, Alignment.cpp:131-134);
 * out_cost[n_utts] the returned path cost. */
SR_API int sr_align_corpus(sr_model* m, sr_corpus* c, const uint16_t* automata, const uint64_t* aut_off,
                    const double tdp[3], uint16_t silence_state, int gmm_kernel,
                    uint16_t* out_states, double* out_cost);

/* Aligner::align_sequence_pruned (Alignment.cpp:149-288): beam `pruning_threshold` on the per-frame
 * best, transition penalty keyed on the destination state, ends at the highest position reached. */
SR_API int sr_align_corpus_pruned(sr_model* m, sr_corpus* c, const uint16_t* automata, const uint64_t* aut_off,
                           const double tdp[3], uint16_t silence_state, double pruning_threshold, int gmm_kernel,
                           uint16_t* out_states, double* out_cost);

/* ---- training-side caller of the scorer: Trainer::calc_am_score (Training.cpp:585-612) -----------------------
 * out[t] = MixtureModel::score(frame t, states[t]) for every frame of the corpus, i.e. the emission cost along
 * a given state path (an alignment from sr_align_corpus*).  The reference's average AM score is the sequential
 * sum of out[] divided by the frame count; the sum is left to the host so that it keeps the reference's order. */
SR_API int sr_path_scores_corpus(sr_model* m, sr_corpus* c, const uint16_t* states, int gmm_kernel, double* out);

/* ---- EM statistics: MixtureModel::accumulate (Mixtures.cpp:278-372) after reset_accumulators (:235-247) --------
 * Tying: accumulator row of every density (mixture order), as MixtureDensity{mean_idx, var_idx} (Types.hpp:19-27).
 * sr_model_load_mixset installs the file's own; models from sr_model_create default to one row per density. */
SR_API int sr_model_set_tying(sr_model* m, uint32_t n_mean, uint32_t n_var, const uint32_t* dens_mean, const uint32_t* dens_var);
SR_API int sr_model_tying_info(const sr_model* m, uint32_t* n_mean, uint32_t* n_var);
/* The topology a trainer needs to turn sr_accumulate_corpus' statistics into a MIXSET file (sr_mixset_write) or the next model
 * (sr_model_create_from_statistics): dens_off[n_states + 1], and the accumulator rows dens_mean[C], dens_var[C] of every density
 * in mixture order (MixtureModel::mixtures_, Mixtures.hpp:88).  Any pointer may be NULL. */
SR_API int sr_model_topology(const sr_model* m, uint32_t* dens_off, uint32_t* dens_mean, uint32_t* dens_var);
/* states[total_frames]: aligned mixture per frame (an alignment from sr_align_corpus*).  first_pass: density 0 of
 * the mixture gets every frame; else max_approx: the arg-min density; else soft memberships.  Outputs (host):
 * mean_acc[n_mean*dim], mean_w[n_mean], var_acc[n_var*dim] (starts at 1e-4 like the reference), var_w[n_var].
 * Rows are summed in frame order, so max-approx / first-pass results are bit-identical to the reference's; with
 * several GPUs every rank accumulates its shard and the four arrays are all-reduced (sum) by the caller. */
SR_API int sr_accumulate_corpus(sr_model* m, sr_corpus* c, const uint16_t* states, int first_pass, int max_approx,
                                double* mean_acc, double* mean_w, double* var_acc, double* var_w);
/* The single-device EM iteration without the PCIe round trip: call sr_accumulate_corpus with all four output arrays NULL -- the
 * statistics then stay in the corpus handle on the device -- and finalise them here into the next model (MixtureModel::finalize,
 * Mixtures.cpp:374-461; same topology and tying as `m`).  The tables are built in HBM; only the 2 x n weights and the variances
 * (for the host-side logarithms, see sr_model_create_from_statistics) cross the bus. */
SR_API int sr_model_create_from_accumulated(sr_model* m, sr_corpus* c, int pooling, int max_approx, sr_model** out);

/* ---- bigram-LM beam search over a linear lexicon ---------------------------------------------------------------
 * Replaces Teaching::LinearSearch (rwth-asr-0.5/src/Teaching/LinearSearch.cc: initialize :489-495, processFrame
 * :496-515, getResult :517-520) for a whole corpus.  Scores are float there (Teaching/Types.hh:17); the acoustic
 * score of mixture m at frame t is (float) of this library's FP64 score.
 *   word_off[W+1], mixtures[]: the linear lexicon (LinearSearch::buildLinearLexicon :477-483), mixture = emission state;
 *   lm[w*W + h] = -log p(w | h) (SearchInterface::getLanguageModelScore(w, h), SearchInterface.cc:77-81);
 *   tdp[8] = [isSilence][loop, forward, skip, exit] (SearchSpace::setTransitionScores :169-180).
 * PARITY UNPINNED: checked against the CPU restatement under oracle/ only (the toolkit cannot be built here). */
typedef struct sr_bigram sr_bigram;
typedef struct {
  float acoustic_pruning;   /* "acoustic-pruning" (:441-442); >= FLT_MAX: off */
  float lm_pruning;         /* "lm-pruning" (:444-445) */
  int gmm_kernel;
  uint32_t max_word_ends;   /* traceback book capacity per frame and utterance; 0 = W (cannot overflow) */
  int flags;                /* 0, or SR_BIGRAM_DENSE_STATES */
} sr_bigram_params;
/* Lexica whose words all have at most four states (and at most 3072 words) keep the state hypotheses of a word in the registers
 * of one lane (viterbi_bigram.hip, KS > 0); this flag keeps them in the dense LDS image every other lexicon uses.  Same results;
 * for cross-checking the two. */
#define SR_BIGRAM_DENSE_STATES 1
SR_API int sr_bigram_create(sr_model* m, uint32_t n_words, const uint32_t* word_off, const uint16_t* mixtures,
                            uint32_t silence_word, const float* lm, const float tdp[8], sr_bigram** out);
SR_API int sr_bigram_destroy(sr_bigram* b);
/* out_word/out_score/out_time: capacity n_frames + n_utts (LinearSearch::getResult's traceback items, silence included);
 * out_off[n_utts+1]: items of utterance u are [out_off[u], out_off[u+1]).  SR_ELIMIT if a book overflowed. */
SR_API int sr_recognize_bigram_corpus(sr_model* m, sr_corpus* c, sr_bigram* b, const sr_bigram_params* p,
                                      uint32_t* out_word, float* out_score, uint32_t* out_time, uint64_t* out_off);

/* Diagnostic: does the fp16 matrix pipe keep subnormal inputs on this device (an assumption of SR_GMM_PREFILTER's
 * error bound; when it does not hold, models are scored by SR_GMM_EXACT's kernel instead)? */
SR_API int sr_probe_fp16_denormals(int device, int* preserved);
/* ... and does its fp32 accumulation stay inside the bound's model?  Adversarial 96-term (dimension <= 46) and 128-term
 * (dimension 47..62) fp16 dot products with known exact sums through the prefilter's own MFMA chains: within_model = 1 iff every
 * |error| <= 87 (96 terms) / 132 (128 terms) * 2^-24 * sum |a_k b_k|; worst_ratio (may be NULL) = the largest
 * |error| / (2^-24 * sum |a_k b_k|) seen, the 128-term ratios scaled by 87 / 132.  The probes run at model creation (the chain
 * length the model uses). */
SR_API int sr_probe_fp16_accumulation(int device, int* within_model, double* worst_ratio);

/* ---- measurement --------------------------------------------------------------------------------
 * When enabled, every kernel launch of this model handle is bracketed by HIP events on the
 * launch stream; sr_profile_read() synchronises and returns accumulated device times. */
typedef struct {
  double gmm_ms;        /* GMM scoring kernel(s) */
  uint64_t gmm_launches;
  double gmm_flops;     /* algorithmic: 4 * dim * densities * frames per launch, summed */
  double search_ms;     /* Viterbi decode / align kernels */
  uint64_t search_launches;
  double search_bytes;  /* algorithmic: (8*S + 4*P) * frames (decode) / (8+1)*N * frames (align) */
  uint64_t frames;      /* frames processed */
  uint64_t refined_pairs;      /* SR_GMM_PREFILTER: (frame, state) pairs scored ... */
  uint64_t refined_densities;  /* ... and densities the FP64 stage had to evaluate for them (>= 1 per pair) */
  double prefilter_ms;         /* SR_GMM_PREFILTER: the 16-bit MFMA pass (with the feature transpose) ... */
  double refine_ms;            /* ... and the FP64 refinement; both are inside gmm_ms */
} sr_profile;
SR_API int sr_profile_enable(sr_model* m, int on);
SR_API int sr_profile_reset(sr_model* m);
SR_API int sr_profile_read(sr_model* m, sr_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* SRGPU_H */
