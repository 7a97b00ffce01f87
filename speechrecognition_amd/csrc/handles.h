// handles.h -- the opaque handle types of the C ABI (include/srgpu.h) and the small helpers the translation units that
// implement it share (srgpu_api.cpp: compute entry points; feeder.cpp: frame-batch feeder and multi-device driver).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srgpu.h"
#include "host_util.h"
#include "kernels.h"

namespace srhost {
// stores a printf-style message for sr_last_error() (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace srhost

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return srhost::fail(SR_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; n = 0; }
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const T* src, size_t count) {
    hipError_t e = ensure(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void swap(DevBuf& o) { T* tp = p; p = o.p; o.p = tp; size_t tn = n; n = o.n; o.n = tn; }
  // handles own their buffers: whatever a destroy function does not release by name goes with the object
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
};

struct EventPair { hipEvent_t a, b; int kind; };  // kind 0 = gmm, 1 = search, 2 = prefilter pass, 3 = refinement (inside 0)


struct sr_feeder;  // feeder.cpp: the asynchronous upload of a corpus


// The device buffers of a destroyed corpus that its model keeps for the next one (sr_recognize_batch creates and destroys a corpus
// per call: nine hipMalloc / hipFree pairs, about a millisecond of the 1.5 ms a boundary step used to lose against a resident one)
struct CorpusSpare {
  DevBuf<float> feats;
  DevBuf<uint64_t> d_frame_off;
  DevBuf<uint32_t> utt_order, out_words, out_count, out_flags;
  DevBuf<double> tb_score;
  DevBuf<uint16_t> tb_word, tb_bkp;
  size_t bytes() const {
    return feats.n * 4 + d_frame_off.n * 8 + (utt_order.n + out_words.n + out_count.n + out_flags.n) * 4 + tb_score.n * 8 + (tb_word.n + tb_bkp.n) * 2;
  }
};
// ... kept only up to this size (SRGPU_SPARE_MB, default 256): a one-shot call on a huge corpus must not leave the model holding
// a corpus-sized allocation; sr_model_trim() releases it at any time
size_t corpus_spare_cap_bytes();

struct sr_model {
  int device = 0;
  uint32_t dim = 0, n_states = 0, ld = 0;
  uint64_t n_dens = 0;
  bool max_approx = true;
  // exact-kernel tables (finalised, per density)
  DevBuf<uint32_t> dens_off;
  DevBuf<double> means, inv_vars, norm, logw;
  // EM tying (accumulator rows)
  DevBuf<uint32_t> dens_mean, dens_var;
  std::vector<uint32_t> h_dens_off, h_dens_mean, h_dens_var;
  // host copies of the finalised tables: the kernel-specific packings are built on first use of that kernel
  std::vector<double> h_means, h_inv_vars, h_norm, h_logw;
  bool mfma_packed = false, pf_packed = false;
  uint32_t n_mean = 0, n_var = 0;
  // MFMA packing
  int ksteps = 0;
  uint32_t n_blocks = 0, n_groups = 0;
  DevBuf<double> apack;
  DevBuf<uint32_t> blk_meta, grp_state;
  std::vector<uint32_t> group_first_block;  // host: [n_groups+1]
  std::map<uint32_t, std::unique_ptr<DevBuf<uint32_t>>> split_tabs, pf_split_tabs;  // state-range split tables by split count
  const uint32_t *split_cur = nullptr, *pf_split_cur = nullptr;
  uint32_t split_ny = 0;
  // fp16 prefilter + FP64 refinement (gmm_prefilter.hip); pf_ks32 == 0: model not eligible
  int pf_ks32 = 0;
  uint32_t pf_groups = 0, pf_ny = 0, max_dens = 0, pf_slots = 0, pf_chunks = 1, pf_pstates = 0;
  DevBuf<unsigned char> pf_apack;
  DevBuf<float> pf_anorm, featsT, featsP;  // featsP: row-major features in the refinement's padded order (only when pf_dp != dim)
  DevBuf<uint4> pf_defer;                  // deferred leftovers of the refinement (mixtures of more than 32 densities): segments ...
  DevBuf<uint32_t> pf_defer_cnt;           // ... and their counts (gmm_refine_defer_layout)
  int neg_possible = -1;                   // can an emission cost of this model be negative?  -1: not looked at yet (srhost::may_go_negative)
  uint32_t pf_dp = 0;                      // gmm_refine_padded_dim(dim): the odd dimension the refinement planes / featsT are laid out in
  DevBuf<uint32_t> pf_mask, pf_ndens, pf_ring;
  DevBuf<double> pf_rows;
  DevBuf<unsigned long long> pf_counter;
  // feeder resources (feeder.cpp), created by the first asynchronous upload and kept: pinning 2 x 8 MiB costs milliseconds
  hipStream_t s_copy = nullptr;
  float* staging[2] = {nullptr, nullptr};
  hipEvent_t staging_free[2] = {nullptr, nullptr};
  std::atomic<bool> staging_busy{false};
  std::mutex spare_mu;
  std::unique_ptr<CorpusSpare> spare;  // of the last corpus destroyed while none was kept
  std::vector<sr_corpus*> corpora;     // live corpora of this model (under spare_mu): sr_model_destroy clears their `model` links, so that a
                                       // corpus destroyed AFTER its model -- against srgpu.h -- frees its buffers without touching the freed model
  // streams / workspace
  hipStream_t s_gmm = nullptr, s_search = nullptr;
  DevBuf<double> scores[2];
  hipEvent_t ev_scored[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  size_t chunk_frames = 0;
  bool overlap = true;  // search of chunk i on its own stream while chunk i+1 is scored (SRGPU_OVERLAP=0: one stream)
  // profiling
  bool profiling = false;
  std::vector<EventPair> events;
  sr_profile prof{};
};

struct sr_corpus {
  sr_model* model = nullptr;
  sr_feeder* feeder = nullptr;      // sr_corpus_upload_async: pieces still on their way (feeder.cpp)
  uint32_t n_utts = 0;
  uint64_t n_frames = 0;
  std::vector<uint64_t> frame_off;  // host copy
  DevBuf<float> feats;
  DevBuf<uint64_t> d_frame_off;
  DevBuf<uint32_t> utt_order;       // per launch range (chunk), its utterances longest first; built for order_chunk_frames
  size_t order_chunk_frames = 0;    // 0 = not built
  // search outputs (device)
  DevBuf<double> tb_score;
  DevBuf<uint16_t> tb_word, tb_bkp;
  DevBuf<uint32_t> out_words, out_count, out_flags;
  DevBuf<unsigned char> big_ws;     // decode_big_kernel: hypothesis arrays of the utterances in flight
  // aligner workspace
  DevBuf<uint16_t> automata, out_states;
  DevBuf<uint64_t> aut_off, bp_off, al_blk_frame0;
  DevBuf<uint32_t> al_list_off, al_states, al_blk_frames, al_blk_list;
  DevBuf<uint8_t> backptr;
  DevBuf<double> out_cost, path_scores;
  // EM accumulation workspace
  DevBuf<uint64_t> pair_off;
  DevBuf<uint32_t> pair_frame, key_mean, key_var, iota, keys_sorted, pairs_sorted, row_begin;
  DevBuf<double> pair_w, acc_mean, acc_var, w_mean, w_var;
  bool acc_valid = false;           // acc_* hold the statistics of the last sr_accumulate_corpus (for sr_model_create_from_accumulated)
  uint32_t acc_n_mean = 0, acc_n_var = 0;
  DevBuf<unsigned char> sort_temp;
};

struct sr_lexicon {
  sr_model* model = nullptr;
  uint32_t n_words = 0, n_slots = 0, silence_idx = 0, silence_state = 0;
  double tdp[3] = {0, 0, 0};
  DevBuf<uint32_t> slot_info, slot_word, word_end_slot;
  // type-sorted copy for the fast kernel
  DevBuf<uint32_t> f_state, f_pred, f_orig, f_type, f_word;
  uint32_t f_n = 0, f_init = 0, f_init_end = 0;
  DevBuf<uint32_t> w_info;   // word-per-lane network (viterbi_words.hip); empty unless every word has <= 4 positions
  DevBuf<uint2> w_states;
  DevBuf<uint32_t> w_order;
  uint32_t w_plain_len = 0;  // 0: no word-per-lane network
  uint32_t w_nw = 0, w_nt = 0, w_general = 0;
  bool big = false;                 // more slots than the LDS kernels hold: decode_big_kernel, no type-sorted copy
};

struct sr_bigram {
  sr_model* model = nullptr;
  uint32_t n_words = 0, silence = 0, n_positions = 0, max_slot_states = 0, silence_states = 0;
  uint32_t row4_mask = 0;  // bit k: some word of slot row k (words k * 1024 .. k * 1024 + 1023) has four states (register layout)
  float tdp[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  DevBuf<uint32_t> slot_off, slot_mix;
  DevBuf<uint16_t> mixtures;
  DevBuf<uint32_t> pos_info, pos_slot;
  DevBuf<float> lmT, lm_rowmin, lm_rowmax;
  // workspace
  DevBuf<uint32_t> we_slot, we_bp, out_word, out_time, out_count, out_flags;
  DevBuf<float> we_score, out_score;
  DevBuf<uint4> book;
  DevBuf<uint64_t> book_off;
};

namespace srhost {
// (srgpu_api.cpp) handle without parameter tables / lazily fetched host copies of them
int model_shell(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, int max_approx, sr_model** out);
int ensure_host_tables(sr_model* m);
int may_go_negative(sr_model* m, bool* out);  // can an emission cost of this model be negative or NaN? (cached)
// (em_finalize.hip) MixtureModel::finalize on the device: statistics (host arrays) -> new model whose tables are built in HBM
int finalize_on_device(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean, uint32_t n_var,
                       const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc, const double* mean_w,
                       const double* var_acc, const double* var_w, int pooling, int max_approx, sr_model** out);
int finalize_accumulated(sr_model* m, sr_corpus* c, int pooling, int max_approx, sr_model** out);
// (feeder.cpp) make frames [f0, f1) of the corpus visible to work queued on `stream` afterwards: returns at once for a
// synchronously uploaded corpus; for sr_corpus_upload_async it waits (host) until the feeder has issued the pieces that
// cover the range and makes the stream wait for their copy events.  Returns the feeder's error, if any.
int corpus_ready(sr_corpus* c, uint64_t f0, uint64_t f1, hipStream_t stream);
bool corpus_upload_in_flight(const sr_corpus* c);
void feeder_join(sr_corpus* c);  // blocks until the feeder thread (if any) has finished, frees its staging buffers
// (srgpu_api.cpp) a new corpus takes over the buffers its model kept from the last destroyed one / a dying corpus leaves them
void corpus_register(sr_corpus* c);     // after c->model is set
void corpus_adopt_spare(sr_corpus* c);
void corpus_donate_spare(sr_corpus* c);
}  // namespace srhost
