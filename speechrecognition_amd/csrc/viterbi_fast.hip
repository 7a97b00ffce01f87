// viterbi_fast.hip -- the fast variant of the beam Viterbi decoder (Recognizer::recognizeSequence_pruned,
// sietill/Recognizer.cpp:103-232); see viterbi_decode.hip for the algorithm, the exactness argument and the
// general (replay) kernel.  Same results bit for bit; what differs is the work layout:
//
//   * slots are SORTED BY TYPE on the host (srgpu_api.cpp: build_fast_net) -- entry position 0 / entry position 1
//     / middle / word end, times the silence flags that select the transition penalties and the word penalty --
//     and every type is padded to whole waves.  A wave therefore handles one type: each per-slot decision of
//     the general kernel becomes a wave-uniform branch, a slot evaluates only the candidates its type has, and
//     penalties are scalars.  Predecessors are explicit slot ids (the sort breaks p-1 / p-2 adjacency).
//   * it assumes every emission cost of the frame is >= 0, which makes the reference's pre-AM early-out
//     (Recognizer.cpp:143,173) inert so a slot is the first minimum over its candidates in source order; the
//     first negative cost raises out_flags bit 1 and the workgroup stops: decode_kernel<.., REPLAY=true>
//     then redoes that utterance exactly.
//   * wave reductions use DPP row operations instead of LDS permutes.
//
// Ties are broken by ORIGINAL slot index (the reference's visiting order), which every slot carries along.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "traceback.h"

namespace srgpu {

static constexpr double kInfF = __builtin_huge_val();

// ---- DPP helpers --------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);  // lanes without a source keep v
}
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_d(double v) {
  return __hiloint2double(dpp_i<CTRL, ROW_MASK>(__double2hiint(v)), dpp_i<CTRL, ROW_MASK>(__double2loint(v)));
}
__device__ inline double readlane63_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// v_min_f64 directly: one instruction per reduction step instead of compare + two selects (fmin() would canonicalise
// its operands first).  Hypothesis scores are never NaN; +inf is an ordinary operand.
__device__ inline double dmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// full-wave minimum, returned to every lane
__device__ inline double wave_min_dpp(double v) {
  v = dmin(v, dpp_d<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
  v = dmin(v, dpp_d<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
  v = dmin(v, dpp_d<0x141, 0xF>(v));   // row_half_mirror
  v = dmin(v, dpp_d<0x140, 0xF>(v));   // row_mirror
  v = dmin(v, dpp_d<0x142, 0xA>(v));   // row_bcast15 -> rows 1, 3
  v = dmin(v, dpp_d<0x143, 0xC>(v));   // row_bcast31 -> rows 2, 3
  return readlane63_d(v);
}
__device__ inline uint32_t wave_min_u32_dpp(uint32_t v) {
  v = umin(v, (uint32_t)dpp_i<0xB1, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x4E, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x141, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x140, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x142, 0xA>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x143, 0xC>((int)v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// full-wave lexicographic (value, index) minimum, returned to every lane: the minimum value first, then the smallest
// index among the lanes that hold it
__device__ inline void wave_min_idx_dpp(double& v, uint32_t& idx) {
  const double m = wave_min_dpp(v);
  idx = wave_min_u32_dpp(v == m ? idx : 0xFFFFFFFFu);
  v = m;
}
// the same over a row of 16 lanes, result in every lane of the row (the cross-wave stage: each row holds all <= 16 partials)
__device__ inline double row_min_dpp(double v) {
  v = dmin(v, dpp_d<0xB1, 0xF>(v));
  v = dmin(v, dpp_d<0x4E, 0xF>(v));
  v = dmin(v, dpp_d<0x141, 0xF>(v));
  v = dmin(v, dpp_d<0x140, 0xF>(v));
  return v;
}
__device__ inline void row_min_idx_dpp(double& v, uint32_t& idx) {
  const double m = row_min_dpp(v);
  uint32_t c = v == m ? idx : 0xFFFFFFFFu;
  c = umin(c, (uint32_t)dpp_i<0xB1, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x4E, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x141, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x140, 0xF>((int)c));
  idx = c;
  v = m;
}

// chunk type word (one per 64 slots): kind | flags
static constexpr uint32_t kKindMask = 7u;
static constexpr uint32_t kE0 = 0, kE0S = 1, kE1 = 2, kE1E = 3, kM = 4, kME = 5, kPad = 6;
static constexpr uint32_t kTSilState = 8u, kTSilWord = 16u, kTFirstSil = 32u;

template <uint32_t K> struct KindC { __device__ constexpr operator uint32_t() const { return K; } };  // a slot kind known at compile time
struct KindR { uint32_t v; __device__ operator uint32_t() const { return v; } };                          // ... or only at run time

template <int NT, int SPT>
__global__ __launch_bounds__(NT) void decode_fast_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t PP = NT * SPT, NW = NT / 64;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* sc = reinterpret_cast<double*>(smem);                  // [PP] hypothesis scores
  double* am_l = sc + PP;                                        // [PP] emission cost of the current frame per slot
  double* red_best = am_l + PP;                                  // [16]
  double* red_we = red_best + 16;                                // [16]
  uint32_t* red_idx = reinterpret_cast<uint32_t*>(red_we + 16);  // [16]
  uint32_t* e_first = red_idx + 16;                              // [2][4] first word-end ORIGINAL index per class, by frame parity
  uint32_t* bail = e_first + 8;                                  // [1]
  uint16_t* bk = reinterpret_cast<uint16_t*>(e_first + 12);      // [PP] back pointers

  const FastNet& net = a.fast;
  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x;
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;

  // Slot of (thread, i): a wave owns SPT consecutive 64-slot chunks of the type-sorted net, so most waves hold one kind (the
  // straight-line path below) and only the waves that hold word-end kinds pay for the word-end reduction.  (Round 1 dealt the
  // chunks round-robin when few utterances were in flight, so that every wave carried the same mix of kinds; with the
  // straight-line path the consecutive deal is faster there too: 2.04 -> 1.87 us per frame for one utterance of configs[1],
  // 1.68 -> 1.49 ms for 64 utterances of configs[2].)
  auto slot_of = [&](int i) -> uint32_t { return (wave * SPT + (uint32_t)i) * 64 + lane; };
  // ---- static per-slot constants ------------------------------------------------------------------------
  uint32_t st[SPT], pr[SPT], og[SPT], ty[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = slot_of(i);
    const bool in = p < net.n_slots;
    st[i] = in ? net.state[p] : 0u;      // padding slots read state 0: a valid address, value unused
    pr[i] = in ? net.pred[p] : (p | (p << 16));
    og[i] = in ? net.orig[p] : 0xFFFFFFFFu;
    ty[i] = __builtin_amdgcn_readfirstlane(in ? net.chunk_type[p >> 6] : kPad);  // one type per 64-slot chunk
    sc[p] = kInfF; bk[p] = 0;
  }
  bool uniform_kind = true;  // wave-uniform: all of this wave's chunks have the same type word
#pragma unroll
  for (int i = 1; i < SPT; i++) uniform_kind &= ty[i] == ty[0];
  bool wave_has_we = false;  // wave-uniform: does any of this wave's chunks hold word-end slots?
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t kind = ty[i] & kKindMask;
    wave_has_we |= (kind == kE0S || kind == kE1E || kind == kME);
  }
  if (tid < 8) e_first[tid] = 0xFFFFFFFFu;
  if (tid == 8) *bail = 0;
  __syncthreads();
  const bool init_is_end = net.init_is_end;
  double m_we = init_is_end ? 0.0 : kInfF;
  if (tid == 0) {
    sc[net.init_slot] = 0.0;  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
    a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0;
  }
  if (tid < 4 && init_is_end) e_first[4 + tid] = 0;
  double am_n[SPT];  // emission gathers one frame ahead: frame t+1's costs are issued at the top of frame t
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = slot_of(i);
    am_n[i] = T > 0 ? row0[st[i]] : 0.0;
    am_l[p] = am_n[i];
  }
  __syncthreads();

  auto frame = [&](const uint32_t t) -> bool {
    const uint32_t* ef_cur = e_first + 4 * (t & 1);
    uint32_t* ef_nxt = e_first + 4 * ((t + 1) & 1);
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;
    if (t + 1 <= T) {
      const double* rown = row0 + (uint64_t)t * a.ld;  // frame t+1
#pragma unroll
      for (int i = 0; i < SPT; i++) am_n[i] = rown[st[i]];
    }

    // ---- A ----------------------------------------------------------------------------------------------
    double nv[SPT];
    uint32_t nb[SPT];
    double my_best = kInfF, my_we = kInfF;
    uint32_t my_we_idx = 0xFFFFFFFFu;
    bool neg = false;
    // one slot's candidates; `kind` is wave-uniform -- a run-time scalar (KindR) or a compile-time constant (KindC)
    auto slot_a = [&](const auto kind, const uint32_t type, const int i) __attribute__((always_inline)) {
      const uint32_t p = slot_of(i);
      const double am = am_l[p];
      neg |= am < 0.0;
      const bool sil = type & kTSilState;
      const double t_loop = sil ? tf : tl, t_skip = sil ? tf : ts;  // scalars: TdpModel.cpp:19-29 keyed on the destination
      const uint32_t p1 = pr[i] & 0xFFFFu, p2 = pr[i] >> 16;
      double v = kInfF;
      uint32_t src = p;
      if (kind >= kM) {  // middle / word end at position >= 2: skip, forward, [loop]
        v = (sc[p2] + t_skip) + am; src = p2;
        const double n1 = (sc[p1] + tf) + am;
        if (n1 < v) { v = n1; src = p1; }
        if (kind == kM) {
          const double n0 = (sc[p] + t_loop) + am;
          if (n0 < v) { v = n0; src = p; }
        }
        nb[i] = bk[src];
      } else {
        // entry slots: in-word candidates (forward for position 1, loop unless word end) ...
        double am_b = am;
        if (kind == kE1 || kind == kE1E) {
          v = (sc[p1] + tf) + am; src = p1;
          am_b = am_l[p1];  // boundary candidates are scored with position 0's emission (:136,148-151)
        }
        if (kind == kE0 || kind == kE1) {
          const double n0 = (sc[p] + t_loop) + am;
          if (n0 < v) { v = n0; src = p; }
        }
        // ... and the collapsed word-boundary candidate, placed before or after them by source index
        const double wp = (type & kTSilWord) ? 0.0 : wp_word;
        const bool b_skip = (kind == kE1 || kind == kE1E) && !(type & kTFirstSil);
        const double t_b = b_skip ? ts : tf;
        const uint32_t cls = ((type & kTSilWord) ? 0u : 2u) + (b_skip ? 1u : 0u);
        const double n_b = ((m_we + wp) + t_b) + am_b;
        const bool b_first = ef_cur[cls] < (og[i] >> 16);
        const bool take_b = b_first ? !(v < n_b) : (n_b < v);
        const uint32_t in_bkp = bk[src];
        v = take_b ? n_b : v;
        nb[i] = take_b ? bkp_new : in_bkp;
        if (kind == kE0S) {  // one-position word: its dead position-1 slot still feeds best_score (:139,155)
          const double dead = ((m_we + wp) + ((type & kTFirstSil) ? tf : ts)) + am;
          my_best = dead < my_best ? dead : my_best;
        }
      }
      const bool real = og[i] != 0xFFFFFFFFu;  // padding lanes inside a type's last chunk
      v = real ? v : kInfF;
      nv[i] = v;
      my_best = v < my_best ? v : my_best;
      if (kind == kE0S || kind == kE1E || kind == kME) {  // word ends
        const uint32_t o = og[i] & 0xFFFFu;
        if (real && (v < my_we || (v == my_we && o < my_we_idx))) { my_we = v; my_we_idx = o; }
      }
    };
    if (uniform_kind) {
      // All SPT chunks of this wave are of one kind (a wave holds consecutive chunks of the type-sorted net):
      // one scalar branch, then the SPT slots as straight-line code whose LDS reads and FP64 chains interleave -- per slot
      // the loop is a dependent chain, and four waves per SIMD do not hide it
#pragma unroll
      for (int i = 0; i < SPT; i++) { nv[i] = kInfF; nb[i] = 0; }
      switch (ty[0] & kKindMask) {
#define SR_KIND_CASE(K)                                                   \
        case K:                                                           \
          _Pragma("unroll") for (int i = 0; i < SPT; i++) slot_a(KindC<K>{}, ty[0], i); \
          break;
        SR_KIND_CASE(kE0) SR_KIND_CASE(kE0S) SR_KIND_CASE(kE1) SR_KIND_CASE(kE1E) SR_KIND_CASE(kM) SR_KIND_CASE(kME)
#undef SR_KIND_CASE
        default: break;  // kPad
      }
    } else {
#pragma unroll
      for (int i = 0; i < SPT; i++) {
        const uint32_t type = ty[i], kind = type & kKindMask;
        nv[i] = kInfF; nb[i] = 0;
        if (kind == kPad) continue;  // wave-uniform
        slot_a(KindR{kind}, type, i);
      }
    }
    if (neg) *bail = 1;
    if (tid < 4) ef_nxt[tid] = 0xFFFFFFFFu;

    // ---- B ----------------------------------------------------------------------------------------------
    my_best = wave_min_dpp(my_best);
    if (wave_has_we) wave_min_idx_dpp(my_we, my_we_idx);  // (other waves keep +inf / no index)
    if (lane == 0) { red_best[wave] = my_best; red_we[wave] = my_we; red_idx[wave] = my_we_idx; }
    __syncthreads();
    if (*bail) {  // workgroup-uniform: hand the utterance to the replay variant
      if (tid == 0) { atomicOr(&a.out_flags[u], kFlagReplay); a.out_count[u] = 0; }
      return true;
    }
    double best = red_best[lane & (NW - 1)], we = red_we[lane & (NW - 1)];
    uint32_t we_idx = red_idx[lane & (NW - 1)];
    if (NW > 1) {  // NW <= 16 partials, replicated in every row of 16 lanes: a row-level reduction is enough
      static_assert(NW <= 16, "one row of 16 lanes holds all per-wave partials");
      best = row_min_dpp(best);
      row_min_idx_dpp(we, we_idx);
    }

    // ---- C ----------------------------------------------------------------------------------------------
    const double limit = best + thr;
    const bool we_alive = !(we > limit) && we != kInfF;
    m_we = we_alive ? we : kInfF;
    const double near = m_we + (fabs(m_we) + fabs(wp_word) + fabs(tf) + fabs(ts) + 1.0) * 1e-9;
#pragma unroll
    for (int i = 0; i < SPT; i++) {  // the stores of all slots first: straight-line
      const uint32_t p = slot_of(i);
      double v = nv[i];
      if (v > limit) v = kInfF;  // :194-196
      nv[i] = v;
      sc[p] = v;
      bk[p] = (uint16_t)nb[i];
      am_l[p] = am_n[i];
    }
    if (wave_has_we && we_alive) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < SPT; i++) {
        const uint32_t kind = ty[i] & kKindMask;
        if (kind == kE0S || kind == kE1E || kind == kME) {
          const double v = nv[i];
          if (v <= near) {
            const uint32_t p = slot_of(i), o = og[i] & 0xFFFFu;
            if (o == we_idx) { a.tb_score[tb0 + t] = v; a.tb_word[tb0 + t] = (uint16_t)p; a.tb_bkp[tb0 + t] = (uint16_t)nb[i]; }
            if (v + 0.0 + tf == m_we + 0.0 + tf) atomicMin(&ef_nxt[0], o);
            if (v + 0.0 + ts == m_we + 0.0 + ts) atomicMin(&ef_nxt[1], o);
            if (v + wp_word + tf == m_we + wp_word + tf) atomicMin(&ef_nxt[2], o);
            if (v + wp_word + ts == m_we + wp_word + ts) atomicMin(&ef_nxt[3], o);
          }
        }
      }
    }
    if (!we_alive && tid == 0) { a.tb_score[tb0 + t] = kInfF; a.tb_word[tb0 + t] = 0xFFFFu; a.tb_bkp[tb0 + t] = 0; }
    __syncthreads();
    return false;
  };

  for (uint32_t t = 1; t <= T; t++)
    if (frame(t)) return;

  // ---- traceback (Recognizer.cpp:222-231; guarded walk: traceback.h) -------------------------------------------
  __threadfence();
  __syncthreads();
  bool bad = false;
  for (uint32_t t = 1 + tid; t <= T; t += NT) {  // winning slot -> word
    const uint32_t sl = __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint16_t w = 0;  // 0xFFFF: no surviving word end -> Book(inf, 0, 0), :118,191
    if (sl != 0xFFFFu) { if (sl < net.n_slots) w = (uint16_t)net.word[sl]; else bad = true; }
    a.tb_word[tb0 + t] = w;
  }
  if (bad) atomicOr(&a.out_flags[u], kFlagCorrupt);
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const uint32_t n = walk_traceback(
        T, a.net.silence_word, a.net.n_words,
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_bkp[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        a.out_words + f0, T);
    if (n == kTbCorrupt) atomicOr(&a.out_flags[u], kFlagCorrupt);
    a.out_count[u] = n == kTbCorrupt ? 0u : n;
  }
}

static size_t fast_smem(uint32_t PP) { return (size_t)PP * 16 + 16 * 8 * 2 + 16 * 4 + 12 * 4 + (size_t)PP * 2 + 16; }

hipError_t launch_decode_fast(const DecodeArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const uint32_t P = a.fast.n_slots;
  const dim3 grid(a.n_utts);
#define SR_LAUNCH(NT, SPT)                                                                                              \
  do {                                                                                                                  \
    const size_t smem = fast_smem((NT) * (SPT));                                                                        \
    hipError_t e = hipFuncSetAttribute((const void*)decode_fast_kernel<NT, SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (e != hipSuccess) return e;                                                                                      \
    hipLaunchKernelGGL((decode_fast_kernel<NT, SPT>), grid, dim3(NT), smem, stream, a);                                 \
    return hipGetLastError();                                                                                           \
  } while (0)
  if (a.n_utts <= 256) {
    // At most one utterance per CU: the frame loop is a chain of dependent steps, and the widest workgroup makes it shortest
    // (one 10 000-frame utterance, us per frame: P = 1216: 3.11 at 256 x 8, 2.24 at 512 x 4, 1.98 at 1024 x 2; P = 448: 1.97 at
    // 256 x 4, 1.47 at 1024 x 1; P = 256: 2.13 at 64 x 4, 1.35 at 256 x 1).
    if (P <= 64) SR_LAUNCH(64, 1);
    if (P <= 256) SR_LAUNCH(256, 1);
    if (P <= 1024) SR_LAUNCH(1024, 1);
    if (P <= 2048) SR_LAUNCH(1024, 2);
  } else {  // many utterances per CU over the launch: instruction throughput
    if (P <= 64) SR_LAUNCH(64, 1);
    if (P <= 256) SR_LAUNCH(64, 4);
    if (P <= 1024) SR_LAUNCH(256, 4);
    if (P <= 2048) SR_LAUNCH(512, 4);  // 1000 utterances at P = 1216: 8.31 ms per step against 8.53 (256 x 8) and 9.08 (1024 x 2)
  }
  if (P <= 4096) SR_LAUNCH(1024, 4);  // measured 7.5 ms vs 8.4 ms for 512 x 8 on 1000 utterances of P = 4000
  if (P <= 8192) SR_LAUNCH(1024, 8);
#undef SR_LAUNCH
  return hipErrorInvalidValue;
}

}  // namespace srgpu
