// viterbi_decode.hip -- time-synchronous beam Viterbi over whole-word linear HMMs with a zerogram
// LM, bit-compatible with Recognizer::recognizeSequence_pruned (sietill/Recognizer.cpp:103-232).
//
// One workgroup decodes one utterance.  The reference's hypothesis array (Book[word*max_pos+pos],
// Recognizer.cpp:116-117) becomes P = sum_w n_pos(w) "slots" in the same (word, position) order;
// a slot's word and position are static, so per slot only {score f64, bkp u16} live in LDS and
// each thread keeps its slots' constants in registers.  Per frame:
//
//   A  every slot gathers its candidates -- skip/forward/loop inside the word (keyed on the
//      DESTINATION state, TdpModel.cpp:19-29; word-end slots do not expand, Recognizer.cpp:131) and,
//      for positions 0 and 1, the word-boundary candidate -- and replays the reference's merge
//      (`new > target -> continue; new += am; target > new -> replace`, :143-157, :173-186) in
//      ascending source-slot order, which is the order the reference visits hypotheses in (:126).
//      The O(#word-ends x W) boundary loop (:133-158) collapses to one min over word-end slots
//      plus a broadcast: all boundary candidates of a target write identical fields and FP addition
//      of a constant is monotone, so only their minimum and the FIRST slot index attaining it (per
//      (word-penalty, tdp) class, for exact ties against in-word candidates) matter.  That shortcut
//      needs the pre-AM early-out (:143) to be inert, i.e. emission costs >= 0; a slot whose
//      candidates involve a negative emission cost replays the boundary loop sequentially instead
//      (exact, O(W), rare: flagged in out_flags bit 0).
//   B  block-wide min of the new scores (= best_score, :155,184; it includes the dead
//      position-1 slot of one-position words such as silence) and min over word-end slots.
//   C  prune `score > best + am_threshold` (:194), record traceback[t] = first minimal surviving
//      word end (:199-205), publish the word-end minimum and first-index per class for frame t+1.
//
// HBM traffic per frame is the emission row gather (8 B per slot) plus one 12-byte traceback
// entry; everything else stays in LDS/registers: the kernel is bound by HBM/L2 latency of the
// score rows and two workgroup barriers per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

// slot_info bit layout (built in srgpu_api.cpp: build_decode_net)
static constexpr uint32_t kSlotPos0 = 1u << 16;      // position 0 of its word
static constexpr uint32_t kSlotPos1 = 1u << 17;      // position 1
static constexpr uint32_t kSlotEnd = 1u << 18;       // last position (word end): no in-word expansion
static constexpr uint32_t kSlotSilState = 1u << 19;  // its state is the silence state: tdp is always `forward`
static constexpr uint32_t kSlotSilWord = 1u << 20;   // word is the silence word: no word penalty
static constexpr uint32_t kSlotFirstSil = 1u << 21;  // word's first state is the silence state
static constexpr uint32_t kSlotSingle = 1u << 22;    // one-position word: owns the virtual dead slot (w, 1)

static constexpr double kInf = __builtin_huge_val();

struct Merge {  // one target hypothesis being built (Book, Recognizer.hpp:75-89; word/pos are static)
  double score;
  uint32_t bkp;
  __device__ Merge() : score(kInf), bkp(0) {}
  // Recognizer.cpp:143-157 / :173-186
  __device__ void offer(double pre_am, double am, uint32_t cand_bkp) {
    if (pre_am > score) return;
    const double n = pre_am + am;
    if (score > n) { score = n; bkp = cand_bkp; }
  }
};

__device__ inline double shfl_xor_f64(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

template <int SPT, int NTMAX>
__global__ __launch_bounds__(NTMAX) void decode_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t P = a.net.n_slots, W = a.net.n_words;
  const uint32_t NT = blockDim.x, tid = threadIdx.x;
  const uint32_t n_waves = NT >> 6, wave = tid >> 6, lane = tid & 63;
  double* sc = reinterpret_cast<double*>(smem);                    // [P] hypothesis scores
  double* red_best = sc + P;                                       // [16]
  double* red_we = red_best + 16;                                  // [16]
  uint32_t* red_idx = reinterpret_cast<uint32_t*>(red_we + 16);    // [16]
  uint32_t* e_first = red_idx + 16;                                // [2][4] first word-end slot per class, by frame parity
  uint16_t* bk = reinterpret_cast<uint16_t*>(e_first + 8);         // [P] back pointers (start frame of the word)

  const uint32_t u = a.utt_first + blockIdx.x;
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;  // traceback[0] of this utterance
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;

  // ---- static per-slot constants ---------------------------------------------------------------
  uint32_t info[SPT], first_state[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = tid + i * NT;
    info[i] = p < P ? a.net.slot_info[p] : 0u;
    // emission state of the word's position 0 (boundary candidates are scored with it even when
    // they enter position 1, Recognizer.cpp:136,148-151)
    first_state[i] = (info[i] & kSlotPos1) ? (a.net.slot_info[p - 1] & 0xFFFFu) : (info[i] & 0xFFFFu);
    if (p < P) { sc[p] = kInf; bk[p] = 0; }
  }
  if (tid < 8) e_first[tid] = 0xFFFFFFFFu;
  __syncthreads();
  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
  double m_we = kInf;  // min over live word-end slots of the previous frame
  if (a.net.slot_info[0] & kSlotEnd) m_we = 0.0;
  if (tid == 0) {
    sc[0] = 0.0;
    if (a.net.slot_info[0] & kSlotEnd) { e_first[0] = e_first[1] = e_first[2] = e_first[3] = 0; }  // parity of t=1 is 1 -> buffer 4..7
    a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0;  // traceback[0] = Book(0.0,0,0,0), :118
  }
  if (tid < 4 && (a.net.slot_info[0] & kSlotEnd)) e_first[4 + tid] = 0;
  __syncthreads();
  uint32_t slow_taken = 0;

  for (uint32_t t = 1; t <= T; t++) {
    const double* row = row0 + (uint64_t)(t - 1) * a.ld;
    const uint32_t* ef_cur = e_first + 4 * (t & 1);
    uint32_t* ef_nxt = e_first + 4 * ((t + 1) & 1);
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;  // merge_hypothesis(.., t - 1, ..) truncated to uint16 (:154, Recognizer.hpp:79)

    // ---- A: build the new hypotheses in registers ------------------------------------------------
    double nv[SPT];
    uint32_t nb[SPT];
    double my_best = kInf, my_we = kInf;
    uint32_t my_we_idx = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      const uint32_t p = tid + i * NT;
      nv[i] = kInf; nb[i] = 0;
      if (p >= P) continue;
      const uint32_t inf = info[i];
      const double am = row[inf & 0xFFFFu];
      const bool sil_state = inf & kSlotSilState;
      const double t_loop = sil_state ? tf : tl, t_fwd = tf, t_skip = sil_state ? tf : ts;
      Merge mg;
      if (!(inf & (kSlotPos0 | kSlotPos1))) {
        // position >= 2: skip from p-2, forward from p-1, loop from p (not for a word end)
        mg.offer(sc[p - 2] + t_skip, am, bk[p - 2]);
        mg.offer(sc[p - 1] + t_fwd, am, bk[p - 1]);
        if (!(inf & kSlotEnd)) mg.offer(sc[p] + t_loop, am, bk[p]);
      } else {
        const bool pos1 = inf & kSlotPos1;
        const double am_b = pos1 ? row[first_state[i]] : am;
        const double wp = (inf & kSlotSilWord) ? 0.0 : wp_word;
        // tdp(first_state, init + 1): init 0 -> forward; init 1 -> skip unless the first state is silence
        const bool b_skip = pos1 && !(inf & kSlotFirstSil);
        const double t_b = b_skip ? ts : tf;
        const uint32_t cls = ((inf & kSlotSilWord) ? 0u : 2u) + (b_skip ? 1u : 0u);
        if (am_b >= 0.0 && am >= 0.0) {
          const double c_b = (m_we + wp) + t_b;   // cur_hyp->score + word_penalty + tdp, :140
          const uint32_t e_b = ef_cur[cls];       // first word-end slot attaining c_b
          if (!pos1) {
            if (e_b < p) mg.offer(c_b, am_b, bkp_new);
            if (!(inf & kSlotEnd)) mg.offer(sc[p] + t_loop, am, bk[p]);
            if (e_b >= p) mg.offer(c_b, am_b, bkp_new);
          } else {
            if (e_b < p - 1) mg.offer(c_b, am_b, bkp_new);
            mg.offer(sc[p - 1] + t_fwd, am, bk[p - 1]);
            if (!(inf & kSlotEnd)) mg.offer(sc[p] + t_loop, am, bk[p]);
            if (e_b >= p - 1) mg.offer(c_b, am_b, bkp_new);
          }
        } else {
          // negative emission cost: replay the boundary loop source by source in slot order
          slow_taken = 1;
          const uint32_t base = pos1 ? p - 1 : p;  // slot of position 0 of this word
          uint32_t v = 0;
          for (; v < W; v++) {
            const uint32_t e = a.net.word_end_slot[v];
            if (e >= base) break;
            const double s = sc[e];
            if (s != kInf) mg.offer((s + wp) + t_b, am_b, bkp_new);
          }
          if (pos1) mg.offer(sc[p - 1] + t_fwd, am, bk[p - 1]);
          if (!(inf & kSlotEnd)) mg.offer(sc[p] + t_loop, am, bk[p]);
          for (; v < W; v++) {
            const double s = sc[a.net.word_end_slot[v]];
            if (s != kInf) mg.offer((s + wp) + t_b, am_b, bkp_new);
          }
        }
        if (inf & kSlotSingle) {
          // one-position word: boundary candidates with init = 1 land in a slot past the word's
          // end (Recognizer.cpp:139); it never expands but feeds best_score (:155).
          const bool d_skip = !(inf & kSlotFirstSil);
          const double t_d = d_skip ? ts : tf;
          Merge dead;
          if (am >= 0.0) {
            dead.offer((m_we + wp) + t_d, am, bkp_new);
          } else {
            slow_taken = 1;
            for (uint32_t v = 0; v < W; v++) {
              const double s = sc[a.net.word_end_slot[v]];
              if (s != kInf) dead.offer((s + wp) + t_d, am, bkp_new);
            }
          }
          my_best = dead.score < my_best ? dead.score : my_best;
        }
      }
      nv[i] = mg.score; nb[i] = mg.bkp;
      my_best = mg.score < my_best ? mg.score : my_best;
      if ((inf & kSlotEnd) && (mg.score < my_we || (mg.score == my_we && p < my_we_idx))) { my_we = mg.score; my_we_idx = p; }
    }
    if (tid < 4) ef_nxt[tid] = 0xFFFFFFFFu;

    // ---- B: block reductions -----------------------------------------------------------------------
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double ob = shfl_xor_f64(my_best, m);
      my_best = ob < my_best ? ob : my_best;
      const double ow = shfl_xor_f64(my_we, m);
      const uint32_t oi = __shfl_xor(my_we_idx, m);
      if (ow < my_we || (ow == my_we && oi < my_we_idx)) { my_we = ow; my_we_idx = oi; }
    }
    if (lane == 0) { red_best[wave] = my_best; red_we[wave] = my_we; red_idx[wave] = my_we_idx; }
    __syncthreads();  // also: every read of sc/bk of frame t-1 is done
    double best = red_best[0], we = red_we[0];
    uint32_t we_idx = red_idx[0];
    for (uint32_t w = 1; w < n_waves; w++) {
      const double ob = red_best[w];
      best = ob < best ? ob : best;
      const double ow = red_we[w];
      const uint32_t oi = red_idx[w];
      if (ow < we || (ow == we && oi < we_idx)) { we = ow; we_idx = oi; }
    }

    // ---- C: prune, traceback, publish word-end minimum ------------------------------------------------
    const double limit = best + thr;
    const bool we_alive = !(we > limit) && we != kInf;
    m_we = we_alive ? we : kInf;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      const uint32_t p = tid + i * NT;
      if (p >= P) continue;
      double v = nv[i];
      if (v > limit) v = kInf;  // :194-196
      sc[p] = v;
      bk[p] = (uint16_t)nb[i];
      if ((info[i] & kSlotEnd) && v != kInf && we_alive) {
        if (p == we_idx) {  // first minimal surviving word end -> traceback[t] (:199-205)
          a.tb_score[tb0 + t] = v; a.tb_word[tb0 + t] = (uint16_t)(a.net.slot_word[p]); a.tb_bkp[tb0 + t] = (uint16_t)nb[i];
        }
        // first slot (in index order) whose boundary candidate equals the minimum, per class
        if (v + 0.0 + tf == m_we + 0.0 + tf) atomicMin(&ef_nxt[0], p);
        if (v + 0.0 + ts == m_we + 0.0 + ts) atomicMin(&ef_nxt[1], p);
        if (v + wp_word + tf == m_we + wp_word + tf) atomicMin(&ef_nxt[2], p);
        if (v + wp_word + ts == m_we + wp_word + ts) atomicMin(&ef_nxt[3], p);
      }
    }
    if (!we_alive && tid == 0) { a.tb_score[tb0 + t] = kInf; a.tb_word[tb0 + t] = 0; a.tb_bkp[tb0 + t] = 0; }
    __syncthreads();
  }

  // ---- traceback (Recognizer.cpp:222-231) -------------------------------------------------------------
  __threadfence();
  __syncthreads();
  if (slow_taken) atomicOr(&a.out_flags[u], 1u);
  if (tid == 0) {
    uint32_t* words = a.out_words + f0;
    uint32_t n = 0, t = T;
    while (t > 0) {
      const uint32_t w = __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w != a.net.silence_word) words[n++] = w;
      t = __hip_atomic_load(&a.tb_bkp[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t i = 0; i < n / 2; i++) { const uint32_t x = words[i]; words[i] = words[n - 1 - i]; words[n - 1 - i] = x; }
    a.out_count[u] = n;
  }
}

uint32_t decode_max_slots() { return 8192; }

static size_t decode_smem(uint32_t P) { return (size_t)P * 8 + 16 * 8 * 2 + 16 * 4 + 8 * 4 + (size_t)P * 2 + 16; }

hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const uint32_t P = a.net.n_slots;
  const size_t smem = decode_smem(P);
  const dim3 grid(a.n_utts);
#define SR_LAUNCH(SPT, NT)                                                                                   \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute((const void*)decode_kernel<SPT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((decode_kernel<SPT, NT>), grid, dim3(NT), smem, stream, a);                               \
    return hipGetLastError();                                                                                \
  } while (0)
  if (P <= 64) SR_LAUNCH(1, 64);
  if (P <= 256) SR_LAUNCH(1, 256);
  if (P <= 1024) SR_LAUNCH(4, 256);
  if (P <= 4096) SR_LAUNCH(4, 1024);
  if (P <= 8192) SR_LAUNCH(8, 1024);
#undef SR_LAUNCH
  return hipErrorInvalidValue;
}

}  // namespace srgpu
