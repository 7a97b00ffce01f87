// kernels.h -- argument blocks and launchers of the gfx950 kernels behind libsrgpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srgpu {

// ---- GMM scoring, FP64 MFMA contraction (gmm_mfma.hip) ------------------------------------------
struct GmmMfmaArgs {
  const float* feats;           // [n_frames x dim] float32
  uint64_t n_frames;
  uint32_t dim;
  const double* apack;          // [n_blocks][KSTEPS][64] model rows in MFMA A-fragment order
  const uint32_t* blk_meta;     // [n_blocks] (group << 1) | last-block-of-group
  const uint32_t* grp_state;    // [n_groups*4] state written by slot g of a group, 0xFFFFFFFF = padding
  const uint32_t* split_begin;  // [ny+1] block range of each state-range split (group aligned)
  double* out;                  // [n_frames x ld]
  uint32_t ld;
  uint32_t nx, ny;              // frame tiles x state-range splits
};
int gmm_mfma_ksteps_for_dim(uint32_t dim);   // 0 if dim unsupported
int gmm_mfma_frames_per_tile(int ksteps);
hipError_t launch_gmm_mfma(const GmmMfmaArgs& a, int ksteps, bool sum, hipStream_t stream);

// ---- GMM scoring, direct form, bit-exact with the reference (gmm_exact.hip) ----------------------
struct GmmExactArgs {
  const float* feats;
  uint64_t n_frames;
  uint32_t dim;
  uint32_t n_states;
  const uint32_t* dens_off;  // [n_states+1]
  const double* means;       // [C x dim]
  const double* inv_vars;    // [C x dim]
  const double* norm;        // [C]
  const double* logw;        // [C]
  double* out;               // [n_frames x ld]
  uint32_t ld;
  uint32_t states_per_split; // grid.y splits the state range
};
hipError_t launch_gmm_exact(const GmmExactArgs& a, bool sum, uint32_t n_splits, hipStream_t stream);
// listed mode: only the (frame, state) pairs a forced alignment can touch.  Block b covers blk_frames[b] (<= 256) frames
// of one utterance starting at absolute frame blk_frame0[b] (a.feats = the corpus' first frame) and scores the states
// states[list_off[blk_list[b]] .. list_off[blk_list[b] + 1]) into out[(frame - frame_base) * ld + state].
struct GmmExactList {
  const uint64_t* blk_frame0;
  const uint32_t* blk_frames;
  const uint32_t* blk_list;
  const uint32_t* list_off;
  const uint32_t* states;   // null: dense mode
  uint32_t blk_first;
  uint64_t frame_base;
};
hipError_t launch_gmm_exact_listed(const GmmExactArgs& a, bool sum, const GmmExactList& L, uint32_t n_blocks, hipStream_t stream);
int gmm_exact_frames_per_block();

// ---- exact GMM scoring through an fp16 prefilter (gmm_prefilter.hip) ------------------------------------------------
struct GmmPrefilterArgs {
  const float* feats;
  uint64_t n_frames;
  uint32_t dim;
  const unsigned char* apack;   // [n_groups][8 blocks][KS32][64 lanes][8 fp16], scaled by a power of two
  const float* grp_anorm;       // [n_groups*4][2] largest (scaled) |a|_2 and |konst| over the densities of the state in that slot
  const uint32_t* split_begin;  // [ny+1] group ranges
  uint32_t* mask;               // [group][frame][4 state slots] candidate densities of (frame, state)
  uint32_t nx, ny;
  uint32_t chunks;              // 1, 2 or 4 consecutive state slots (of 32 densities) make up one state
};
struct GmmRefineArgs {
  // dim = the PADDED, odd dimension gmm_refine_padded_dim(model dimension): pair dimensions, zeros, the odd tail last (gmm_prefilter.hip)
  const float* feats;           // [n_frames x dim] row-major features in that order (batches of unrelated frames): the corpus' own
                                // rows when its dimension IS dim, else the padded copy launch_transpose_feats writes
  const float* featsT;          // [dim x n_frames_ld] transposed features (main pass: 64 consecutive frames per wave)
  uint64_t n_frames, n_frames_ld;
  uint32_t dim, n_pstates, chunks;  // pseudo-states = states x chunks (a state of up to 32*chunks densities)
  const uint32_t* n_dens_ps;    // [n_pstates] densities in each pseudo-state (0..32)
  const double* rows;           // [state][2*dim + 2 planes][n_slots]: mu_0, 1/var_0, ..., norm, logw per density slot;
                                // 1 KB of slack after the last state
  uint32_t n_slots;             // gmm_refine_slots(max_dens): 8, 16 or 32
  const uint32_t* mask;         // as written by the prefilter
  double* out; uint32_t ld;
  uint64_t frames_per_split;    // set by the launcher
  unsigned long long* n_refined;  // optional: += densities evaluated (profiling)
  uint32_t* ring;               // workspace, gmm_refine_ring_words() u32: wave-private lists of pairs with further candidates
  // deferred leftovers (mixtures of more than 32 densities, gmm_refine_defer_layout): per wave and panel a segment of defer_cap 16-byte
  // entries + its count; null: the lists are worked off inside the refinement kernel (rounds 2-4)
  void* defer;
  uint32_t* defer_cnt;
  uint32_t defer_cap;
};
// workspace of the deferred-leftover route for this launch: entries per segment (0: the route does not apply -- one chunk per state, a
// padded dimension beyond 39, or more than `budget_bytes`), total 16-byte entries, total counters
void gmm_refine_defer_layout(const GmmRefineArgs& a, size_t budget_bytes, uint32_t* cap, size_t* n_entries, size_t* n_counts);
size_t gmm_refine_ring_words(const GmmRefineArgs& a);  // needs n_frames, n_pstates, dim, n_slots
hipError_t launch_gmm_prefilter(const GmmPrefilterArgs& a, int ks32, hipStream_t stream);
int gmm_prefilter_frames_per_tile();
// does the fp16 MFMA keep subnormal inputs on this device / in this build (an assumption of the prefilter's error bound)?
hipError_t probe_fp16_denormals(hipStream_t stream, bool* preserved);
// does the fp16 MFMA chain accumulate within the bound's model (|error| <= 87 (K = 96: ks32 3) / 132 (K = 128: ks32 4) * 2^-24 *
// sum |a_k b_k| on adversarial dot products)?
hipError_t probe_fp16_accumulation(hipStream_t stream, int ks32, bool* ok, double* worst_ratio);
uint32_t gmm_refine_padded_dim(uint32_t dim);               // odd, >= dim; 0: no refinement instantiation (dim > 62)
int gmm_refine_slots(uint32_t max_dens, uint32_t padded_dim);  // density slots per panel: 8, 16 or 32 (always 32 beyond padded dimension 39)
hipError_t launch_gmm_refine(const GmmRefineArgs& a, hipStream_t stream);
// featsT [dp x ldT] in the padded order and, when dp != dim, featsP [n_frames x dp] row-major in the same order (else unused)
hipError_t launch_transpose_feats(const float* feats, uint64_t n_frames, uint32_t dim, uint32_t dp, uint64_t ldT, float* out, float* featsP,
                                  hipStream_t stream);

// ---- beam Viterbi decoder (viterbi_decode.hip) ---------------------------------------------------
// Search network flattened to "slots" = (word, position) pairs in (word, position) order, which is
// the iteration order of the reference's hypothesis array (Recognizer.cpp:126).
struct DecodeNet {
  uint32_t n_slots;             // P
  uint32_t n_words;
  const uint32_t* slot_info;    // [P] packed per-slot constants, see viterbi_decode.hip
  const uint32_t* slot_word;    // [P] word index of the slot
  const uint32_t* word_end_slot;// [W] slot index of each word's last position
  uint32_t silence_word;
  double tdp_loop, tdp_forward, tdp_skip;
  uint32_t silence_state;
};
// The same network with slots sorted by type for the fast kernel (viterbi_fast.hip); ids below are positions in
// the sorted order, every type padded to a multiple of 64 slots.
struct FastNet {
  uint32_t n_slots;             // padded total (multiple of 64)
  const uint32_t* state;        // [n] emission state of the slot
  const uint32_t* pred;         // [n] sorted id of the slot one position back | two positions back << 16
  const uint32_t* orig;         // [n] original slot index | original index of the word's position 0 << 16; 0xFFFFFFFF = padding
  const uint32_t* chunk_type;   // [n/64] kind | flags (viterbi_fast.hip)
  const uint32_t* word;         // [n] word index of the slot
  uint32_t init_slot;           // sorted id of original slot 0
  uint32_t init_is_end;         // original slot 0 is a word end
};
// The same network word by word for the word-per-lane kernel (viterbi_words.hip): lexica whose words all have <= 4 positions
struct WordNet {
  const uint32_t* info;         // [W] positions (bits 0-2) | silence word (8) | first state is silence (16) | position p's state is silence << (8 + p); null: not built
  const uint2* states;          // [W] emission state of positions 0..3, 16 bits each
  const uint32_t* order;        // [nw * nt] word of lane slot tid + k * nt; every group of 64 slots holds one kind of word, 0xFFFFFFFF = padding
  uint32_t nw, nt;              // words per lane, lanes per workgroup (whole waves)
  uint32_t plain_len;           // positions of a plain word (2, 3 or 4): the commonest length among the words without silence flags
  uint32_t has_general;         // some word is neither plain nor a one-position word
  uint32_t init_is_end;         // word 0 has one position: the initial hypothesis is a word end
};
struct DecodeArgs {
  DecodeNet net;
  FastNet fast;
  WordNet words;
  const double* scores;         // [frames x ld] dense emission costs of this launch's frames
  uint32_t ld;
  const uint64_t* frame_off;    // [n_utts+1] global frame offsets of the corpus
  uint64_t frame_base;          // scores row 0 is global frame `frame_base`
  uint32_t utt_first, n_utts;   // utterances handled by this launch
  const uint32_t* utt_order;    // [n_utts_total] workgroup utt_first + b decodes utterance utt_order[utt_first + b]: the launch's range,
                                // longest first (the tail of a launch is then its shortest utterances); null = identity
  double am_threshold, word_penalty;
  // traceback arrays, entry frame_off[u] + u + t  (t = 0..T_u)
  double* tb_score;
  uint16_t* tb_word;
  uint16_t* tb_bkp;
  uint32_t* out_words;          // utterance u writes its words at out_words[frame_off[u] ...]
  uint32_t* out_count;          // [n_utts_total]
  uint32_t* out_flags;          // [n_utts_total] kFlagSlowPath | kFlagReplay | kFlagCorrupt (traceback.h)
  uint32_t force_general;       // skip the fast kernels: every utterance goes through decode_kernel<.., REPLAY = true>
  uint32_t force_slots;         // the slot-per-lane kernel (viterbi_fast.hip) even where the word-per-lane kernel applies
  uint32_t only_flagged;        // decode_big_kernel as the replay of the word-per-lane kernel: only utterances flagged kFlagReplay
  uint32_t exact_negative;      // the model's emission costs can be negative (some density has norm - log weight < 0): the word-per-lane kernel
                                // runs its NEG variant, which replays the reference's early-out instead of flagging the utterance
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream);       // fast variant, then the replay variant for flagged utterances
hipError_t launch_decode_fast(const DecodeArgs& a, hipStream_t stream);
bool decode_words_applies(const DecodeArgs& a);                          // short-word lexicon whose score rows fit the LDS twice
uint32_t decode_words_max_words();
hipError_t launch_decode_words(const DecodeArgs& a, hipStream_t stream);
uint32_t decode_max_slots();          // what the LDS-resident kernels hold
// lexicons beyond that: hypothesis arrays in a global workspace of n_utts * decode_big_workspace(P) bytes
uint32_t decode_big_max_slots();
size_t decode_big_workspace(uint32_t n_slots);
hipError_t launch_decode_big(const DecodeArgs& a, unsigned char* ws, hipStream_t stream);
// the traceback walk alone (traceback.h) on arrays in the traceback layout above; out_flags[u] = kFlagCorrupt where it does not walk
hipError_t launch_traceback(const uint64_t* frame_off, uint32_t n_utts, const uint16_t* tb_word, const uint16_t* tb_bkp,
                            uint32_t silence_word, uint32_t n_words, uint32_t* out_words, uint32_t* out_count,
                            uint32_t* out_flags, hipStream_t stream);

// ---- forced aligners (viterbi_align.hip) ----------------------------------------------------------
struct AlignArgs {
  const double* scores;         // [frames x ld]
  uint32_t ld;
  const uint64_t* frame_off;    // [n_utts+1]
  uint64_t frame_base;
  uint32_t utt_first, n_utts;
  const uint32_t* utt_order;    // as in DecodeArgs
  const uint16_t* automata;     // concatenated reference automata
  const uint64_t* aut_off;      // [n_utts_total+1]
  double tdp_loop, tdp_forward, tdp_skip;
  uint32_t silence_state;
  double pruning_threshold;     // pruned variant only
  uint8_t* backptr;             // workspace: [sum_u T_u * N_u] taken transition (0/1/2, 3 = none)
  const uint64_t* bp_off;       // [n_utts_total+1] offsets into backptr
  uint32_t max_positions;       // max N_u over the launch (sizes the LDS cost buffers)
  uint16_t* out_states;         // [total frames]
  double* out_cost;             // [n_utts_total]
};
hipError_t launch_align_full(const AlignArgs& a, hipStream_t stream);
hipError_t launch_align_pruned(const AlignArgs& a, hipStream_t stream);
uint32_t align_max_positions();

// ---- bigram-LM beam search over a linear lexicon (viterbi_bigram.hip; Teaching::LinearSearch) -----------------------
struct BigramArgs {
  const double* scores;         // [frames x ld]
  uint32_t ld;
  const uint64_t* frame_off;    // [n_utts_total+1]
  uint64_t frame_base;
  uint32_t utt_first, n_utts;
  const uint32_t* utt_order;    // as in DecodeArgs
  uint32_t n_words, silence, n_positions;  // W; silence word; sum of state counts over the 2W slots (words + silence copies)
  const uint32_t* slot_off;     // [2W+1] first dense position of every slot
  const uint32_t* slot_mix;     // [2W] offset of the slot's acoustic word in `mixtures`
  const uint16_t* mixtures;     // emission state per lexicon position
  const uint32_t* pos_info;     // [n_positions] per dense position (slots back to back): emission state | flags << 16
                                //   flags: 1 = first state of its slot, 2 = second, 4 = last, 8 = silence (copy)
  const uint32_t* pos_slot;     // [n_positions] its slot
  const float* lmT;             // [W x W] transposed: lmT[h*W + w] = -log p(w | h)
  const float *lm_rowmin, *lm_rowmax;  // [W] min / max over w != silence of lmT[h][w]
  float tdp[2][4];              // [isSilence][loop, forward, skip, exit]
  float ac_pruning, lm_pruning; // >= FLT_MAX: off
  uint32_t *we_slot, *we_bp; float* we_score;  // workspace [n_utts][2][2W]
  uint4* book;                  // traceback book; utterance u owns [book_off[u], book_off[u+1])
  const uint64_t* book_off;     // [n_utts_total+1]
  uint32_t* out_word; float* out_score; uint32_t* out_time;  // [frames + utts]: utterance u at frame_off[u] + u
  uint32_t *out_count, *out_flags;  // [n_utts_total]; flag 1 = book capacity exceeded
  uint32_t max_slot_states;     // most states of any slot
  uint32_t silence_states;      // states of the silence word (= of every silence copy)
  uint32_t dense_states;        // keep the state hypotheses in the dense LDS image even where the register layout applies
  uint32_t row4_mask;           // register layout: bit k = slot row k (words k * 1024 ...) holds a word of four states (the other rows keep three)
};
hipError_t launch_bigram(const BigramArgs& a, hipStream_t stream);
size_t bigram_lds_bytes(uint32_t n_words, uint32_t n_positions);
bool bigram_register_layout(const BigramArgs& a);   // short words, <= 3072 of them, the emission row fits the LDS beside the lists
uint32_t bigram_max_words();

// out[f] = scores[(f - frame_base) * ld + states[f]] for f in [f0, f1)  (Trainer::calc_am_score, Training.cpp:605)
hipError_t launch_path_scores(const double* scores, uint32_t ld, uint64_t frame_base, uint64_t f0, uint64_t f1,
                              const uint16_t* states, double* out, hipStream_t stream);

// ---- EM accumulation (em_accumulate.hip) ------------------------------------------------------------------------
struct EmArgs {
  const float* feats;
  uint64_t n_frames, n_pairs;
  uint32_t dim;
  const uint16_t* states;      // [n_frames] aligned mixture per frame
  const uint64_t* pair_off;    // [n_frames] first pair of each frame
  const uint32_t* dens_off;    // model (per density, mixture order)
  const double* means; const double* inv_vars; const double* norm; const double* logw;
  const uint32_t* dens_mean; const uint32_t* dens_var;  // accumulator row of each density
  uint32_t n_mean, n_var;
  int first_pass, max_approx;
  uint32_t* pair_frame; double* pair_w; uint32_t* key_mean; uint32_t* key_var;  // [n_pairs] workspace
};
size_t em_sort_temp_bytes(uint64_t n_pairs);
// out[t] = score(frame t, a.states[t]); uses feats, n_frames, dim, states, dens_off, means, inv_vars, norm, logw, max_approx
hipError_t launch_path_scores_direct(const EmArgs& a, double* out, hipStream_t stream);
hipError_t launch_em_accumulate(const EmArgs& a, void* sort_temp, size_t sort_temp_bytes, uint32_t* iota, uint32_t* keys_sorted,
                                uint32_t* pairs_sorted, uint32_t* row_begin /* [max(n_mean, n_var) + 1] */, double* mean_acc,
                                double* mean_w, double* var_acc, double* var_w, hipStream_t stream);

}  // namespace srgpu
