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
//     (Recognizer.cpp:143,173) inert so a slot is the first minimum over its candidates in source order; an
//     utterance in which any emission cost was negative (or not a number) is flagged after its last frame
//     (out_flags kFlagReplay, no words reported) and decode_kernel<.., REPLAY=true> redoes it exactly.
//   * reductions use DPP row operations and LDS ds_min_f64 cells; the frame's score row is staged in LDS by LDS-DMA.
//
// Ties are broken by ORIGINAL slot index (the reference's visiting order), which every slot carries along.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpp_util.h"
#include "kernels.h"
#include "traceback.h"

namespace srgpu {

// LDS of decode_fast_kernel before the row buffers: scores, the minima cells, first-index cells, flag, back pointers
__host__ __device__ constexpr size_t fast_smem_base(uint32_t PP) { return (size_t)PP * 16 + 4 * 8 + 2 * 4 + 8 * 4 + 2 * 4 + 16; }

// chunk type word (one per 64 slots): kind | flags
static constexpr uint32_t kKindMask = 7u;
static constexpr uint32_t kE0 = 0, kE0S = 1, kE1 = 2, kE1E = 3, kM = 4, kME = 5, kPad = 6;
static constexpr uint32_t kTSilState = 8u, kTSilWord = 16u, kTFirstSil = 32u;

template <uint32_t K> struct KindC { __device__ constexpr operator uint32_t() const { return K; } };  // a slot kind known at compile time
struct KindR { uint32_t v; __device__ operator uint32_t() const { return v; } };                          // ... or only at run time

// One hypothesis in LDS: a 16-byte cell {score, back pointer, -} read and written as ONE 128-bit access -- consecutive lanes,
// consecutive cells: conflict-free, one address register per source.  (Left to the compiler a struct of that shape became a b64
// and a b32 access at a 16-byte stride: 4e8 bank-conflict cycles per launch; two separate arrays were conflict-free but
// doubled the LDS instructions and the hoisted address registers: 412 bytes of scratch in the frame loop.)
struct Cell { double score; uint32_t bkp; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Round 3 ("instruction diet"): the frame loop was 260 vector instructions per wave and frame for 4 slots per
// lane; this version issues 174 (profiles/r3_decoder_diet.txt).  What went:
//   * no LDS copy of the frame's emission costs that every slot wrote and position-1 slots read (the boundary quirk,
//     Recognizer.cpp:148-151: they are scored with position 0's emission): a slot reads its cost -- and position 0's -- where the
//     row is (ROWS, below) or keeps its own one-frame-ahead gathers in registers;
//   * candidates merge as v_min_f64 + one compare + one select of the back pointer (first-wins ties = strict '<' on the later
//     candidate), not compare + three selects; the word-boundary candidate is one add per slot -- (m_we + wp) + tdp is the same
//     in every lane (+inf while no word end is alive) -- and its tie order against the in-word candidates (first word-end
//     index per class vs. the slot's own word) is patched in after the block, only when a wave sees a tie;
//   * "emission cost < 0 or NaN" (the fast path's premise, see the file header) is one compare per slot into a scalar mask
//     that is looked at once, after the last frame, instead of an LDS flag and a workgroup-uniform branch per frame;
//   * the block minimum and the word-end minimum are LDS ds_min_f64 cells (by frame parity) fed by four lanes per wave after a
//     row-level DPP reduction: no partials array, no second reduction after the barrier, no index reduction at all -- the FIRST
//     minimal word end is found by the (one or two) lanes that hold the minimum, through an LDS atomic min on their original
//     index, and the traceback entry of frame t is written at the top of frame t + 1 by the lane that won;
//   * padding lanes cost two selects only in the chunks that have any.
//   * ROWS (round 3, after the diet left the time where it was): the frame's emission costs used to be gathered from the score
//     table by every slot -- 85 wave-wide 8-byte gathers per frame and workgroup at a 24-byte stride, each a dozen cache lines:
//     the waves spent 2 000-3 500 cycles per frame at the top of the loop behind the texture-address queue
//     (profiles/r3_decoder_diet.txt).  Now the whole row of frame t + 1 (states x 8 bytes, contiguous) is copied into LDS by
//     LDS-DMA during frame t -- 1 KB per wave instruction, no registers -- and a slot reads its cost with one ds_read_b64.
//     Rows that do not fit twice beside the hypotheses (S > ~6 000) keep the gathers (ROWS = false).
template <int NT, int SPT, bool ROWS>
__global__ __launch_bounds__(NT) void decode_fast_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t PP = NT * SPT;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // a SCALAR (see viterbi_words.hip: the piece loop below)
  // smem: [PP] hypothesis cells of 16 bytes, then:
  double* c_best = reinterpret_cast<double*>(smem + (size_t)PP * 16);  // [2] block minimum of the frame's new scores, by frame parity
  double* c_we = c_best + 2;                                        // [2] minimum over word-end slots
  uint32_t* c_widx = reinterpret_cast<uint32_t*>(c_we + 2);         // [2] first ORIGINAL index among the minimal word ends
  uint32_t* e_first = c_widx + 2;                                   // [2][4] first word-end original index per class, by frame parity
  uint32_t* s_bad = e_first + 8;                                    // [1] some emission cost was negative or NaN
  // ROWS: two row buffers (frame parity) behind the hypotheses, 1 KB granular (one LDS-DMA piece = 64 lanes x 16 bytes)
  const uint32_t row_bytes = a.ld * 8u, row_pad = (row_bytes + 1023u) & ~1023u;
  unsigned char* rows_lds = smem + ((fast_smem_base(PP) + 1023u) & ~(size_t)1023u);
  auto cell = [&](uint32_t byte_off) -> Cell {  // one ds_read_b128
    const u32x4 r = *reinterpret_cast<const u32x4*>(smem + byte_off);
    return Cell{__hiloint2double((int)r.y, (int)r.x), r.z};
  };
  auto put_cell = [&](uint32_t byte_off, double v, uint32_t b) {  // one ds_write_b128
    *reinterpret_cast<u32x4*>(smem + byte_off) = u32x4{(uint32_t)__double2loint(v), (uint32_t)__double2hiint(v), b, 0u};
  };

  const FastNet& net = a.fast;
  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x;
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;

  // Slot of (thread, i): a wave owns SPT consecutive 64-slot chunks of the type-sorted net, so most waves hold one kind (the
  // straight-line path below) and only the waves that hold word-end kinds pay for the word-end reduction.
  auto slot_of = [&](int i) -> uint32_t { return (wave * SPT + (uint32_t)i) * 64 + lane; };
  // ---- static per-slot constants ------------------------------------------------------------------------
  uint32_t st[SPT], st0[SPT], p1[SPT], p2[SPT], og[SPT], ty[SPT];
  uint32_t pad_chunks = 0;  // wave-uniform: bit i = chunk i of this wave has padding lanes
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = slot_of(i);
    const bool in = p < net.n_slots;
    st[i] = in ? net.state[p] : 0u;      // padding slots read state 0: a valid address, value unused
    const uint32_t pr = in ? net.pred[p] : (p | (p << 16));
    p1[i] = (pr & 0xFFFFu) * 16u; p2[i] = (pr >> 16) * 16u;  // byte offsets of the predecessors' cells
    og[i] = in ? net.orig[p] : 0xFFFFFFFFu;
    ty[i] = __builtin_amdgcn_readfirstlane(in ? net.chunk_type[p >> 6] : kPad);  // one type per 64-slot chunk
    const uint32_t kind = ty[i] & kKindMask;
    st0[i] = (kind == kE1 || kind == kE1E) ? net.state[pr & 0xFFFFu] : st[i];            // emission state of the word's position 0
    if (__any(og[i] == 0xFFFFFFFFu)) pad_chunks |= 1u << i;
    put_cell(p * 16u, kInfF, 0u);
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
  if (tid < 2) { c_best[tid] = kInfF; c_we[tid] = kInfF; c_widx[tid] = 0xFFFFFFFFu; }
  if (tid == 8) *s_bad = 0;
  __syncthreads();
  const bool init_is_end = net.init_is_end;
  double m_we = init_is_end ? 0.0 : kInfF;  // minimum over the word ends that survived the previous frame (uniform)
  if (tid == 0) {
    put_cell(net.init_slot * 16u, 0.0, 0u);  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
    a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0;
  }
  if (tid < 4 && init_is_end) e_first[4 + tid] = 0;
  // emission costs one frame ahead: the row (ROWS) or the gathers of frame t + 1 are issued at the top of frame t
  auto issue_row = [&](uint32_t frame /* 1-based */) {  // row of `frame` -> buffer frame & 1; every wave copies its share of the pieces
    const unsigned char* src = reinterpret_cast<const unsigned char*>(row0 + (uint64_t)(frame - 1) * a.ld);
    unsigned char* dst = rows_lds + (frame & 1u) * row_pad;
    const uint32_t n_full = row_bytes >> 10;
    uint32_t piece = wave;  // (scalar loop; viterbi_words.hip)
    for (; piece < n_full; piece += NT / 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u + lane * 16u),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024u), 16, 0, 0);
    if (piece == n_full && lane * 16u < (row_bytes & 1023u))  // (a row is a multiple of 64 bytes: the last piece may be short)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u + lane * 16u),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024u), 16, 0, 0);
  };
  double am_c[SPT], am0_c[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t kind = ty[i] & kKindMask;
    am_c[i] = (!ROWS && T > 0) ? row0[st[i]] : 0.0;
    am0_c[i] = (!ROWS && T > 0 && (kind == kE1 || kind == kE1E)) ? row0[st0[i]] : am_c[i];
  }
  if (ROWS && T > 0) { issue_row(1); __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)
  __syncthreads();

  uint64_t bad = 0;        // lanes that met an emission cost that is not >= 0 (scalar mask, looked at after the last frame)
  // the lane that may own traceback[t - 1] (it held the word-end minimum of the previous frame): written at the top of frame t
  bool pend = false;
  uint32_t pend_o = 0, pend_p = 0, pend_b = 0;
  double pend_v = 0.0;
  bool prev_alive = true;  // frame 0: traceback[0] is written above
  auto flush_pending = [&](const uint32_t t_prev) {  // t_prev >= 1
    if (__any(pend)) {  // wave-uniform, about one wave per frame
      if (pend && c_widx[t_prev & 1] == pend_o) {
        a.tb_score[tb0 + t_prev] = pend_v; a.tb_word[tb0 + t_prev] = (uint16_t)pend_p; a.tb_bkp[tb0 + t_prev] = (uint16_t)pend_b;
      }
      pend = false;
    }
    if (!prev_alive && tid == 0) { a.tb_score[tb0 + t_prev] = kInfF; a.tb_word[tb0 + t_prev] = 0xFFFFu; a.tb_bkp[tb0 + t_prev] = 0; }
  };

  for (uint32_t t = 1; t <= T; t++) {
    const uint32_t par = t & 1;
    const uint32_t* ef_cur = e_first + 4 * par;
    uint32_t* ef_nxt = e_first + 4 * (par ^ 1);
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;
    double am[SPT], am0[SPT];
#pragma unroll
    for (int i = 0; i < SPT; i++) { am[i] = am_c[i]; am0[i] = am0_c[i]; }
    if (t + 1 <= T) {
      if (ROWS) {
        issue_row(t + 1);
      } else {
        const double* rown = row0 + (uint64_t)t * a.ld;  // frame t + 1
#pragma unroll
        for (int i = 0; i < SPT; i++) {
          const uint32_t kind = ty[i] & kKindMask;
          am_c[i] = rown[st[i]];
          if (kind == kE1 || kind == kE1E) am0_c[i] = rown[st0[i]];
        }
      }
    }
    const double* row_l = reinterpret_cast<const double*>(rows_lds + par * row_pad);  // ROWS: this frame's row
    if (t > 1) flush_pending(t - 1);
    if (tid < 4) ef_nxt[tid] = 0xFFFFFFFFu;
    if (tid == 4) { c_best[par ^ 1] = kInfF; c_we[par ^ 1] = kInfF; c_widx[par] = 0xFFFFFFFFu; }

    // ---- A: candidates of every slot --------------------------------------------------------------------------
    // Staged: every LDS read of the wave's SPT slots is issued before the first value is used, the arithmetic of all slots
    // follows as one block, and the two rare cases (an exact tie between the boundary candidate and an in-word candidate;
    // padding lanes) are patched afterwards -- a per-slot `if (__any(...))` in the middle splits the block, and then each slot
    // pays its own LDS round trip (phase A of the mixed-kind waves was 3 000 cycles of 7 900 per frame that way,
    // profiles/r3_decoder_diet.txt).
    double nv[SPT];
    uint32_t nb[SPT];
    bool tie[SPT];
    uint64_t tie_any = 0;
    double my_best = kInfF, my_we = kInfF;
    const bool we_in = m_we != kInfF;  // uniform: a word end survived the previous frame -- else every boundary candidate is +inf
    struct In { Cell c0, c1, c2; double e, e0; };  // what a slot reads: its sources' cells, its emission cost and position 0's
    // `kind` is wave-uniform -- a run-time scalar (KindR) or a compile-time constant (KindC)
    auto load_slot = [&](const auto kind, const int i, In& in) __attribute__((always_inline)) {
      const bool pos1 = kind == kE1 || kind == kE1E;
      in.e = ROWS ? row_l[st[i]] : am[i];
      in.e0 = pos1 ? (ROWS ? row_l[st0[i]] : am0[i]) : in.e;
      if (kind >= kM) in.c2 = cell(p2[i]);
      if (kind >= kM || pos1) in.c1 = cell(p1[i]);
      if (kind == kM || kind == kE0 || kind == kE1) in.c0 = cell(slot_of(i) * 16u);
    };
    auto compute_slot = [&](const auto kind, const uint32_t type, const int i, const In& in) __attribute__((always_inline)) {
      const double e = in.e;
      bad |= __ballot(!(e >= 0.0));
      const bool sil = type & kTSilState;
      const double t_loop = sil ? tf : tl, t_skip = sil ? tf : ts;  // scalars: TdpModel.cpp:19-29 keyed on the destination
      double v = kInfF;
      uint32_t b = 0;
      tie[i] = false;
      if (kind >= kM) {  // middle / word end at position >= 2: skip, forward, [loop] -- in source order, a later one must be strictly better
        const double s2 = (in.c2.score + t_skip) + e, s1 = (in.c1.score + tf) + e;
        b = s1 < s2 ? in.c1.bkp : in.c2.bkp;
        v = dmin(s2, s1);
        if (kind == kM) {
          const double s0 = (in.c0.score + t_loop) + e;
          b = s0 < v ? in.c0.bkp : b;
          v = dmin(v, s0);
        }
      } else {
        // entry slots: in-word candidates (forward for position 1, loop unless word end) ...
        if (kind == kE1 || kind == kE1E) {
          v = (in.c1.score + tf) + e;
          b = in.c1.bkp;
        }
        if (kind == kE0 || kind == kE1) {
          const double s0 = (in.c0.score + t_loop) + e;
          b = s0 < v ? in.c0.bkp : b;
          v = dmin(v, s0);
        }
        // ... and the collapsed word-boundary candidate (+inf while no word end is alive), placed before or after them by source
        // index: after, unless the patch below finds a tie and an earlier word end.  It is scored with position 0's emission
        // (Recognizer.cpp:136,148-151)
        const double wp = (type & kTSilWord) ? 0.0 : wp_word;
        const bool b_skip = (kind == kE1 || kind == kE1E) && !(type & kTFirstSil);
        const double cb = (m_we + wp) + (b_skip ? ts : tf);  // the same in every lane and for every slot of the class
        const double n_b = cb + in.e0;
        tie[i] = we_in && n_b == v;
        tie_any |= __ballot(tie[i]);
        b = n_b < v ? bkp_new : b;
        v = dmin(v, n_b);
        if (kind == kE0S) {  // one-position word: its dead position-1 slot still feeds best_score (:139,155)
          // (real slots only: a padding lane of this chunk reads state 0, which need not be a one-position word's state)
          const double dead = ((m_we + wp) + ((type & kTFirstSil) ? tf : ts)) + e;
          my_best = dmin(my_best, og[i] != 0xFFFFFFFFu ? dead : kInfF);
        }
      }
      nv[i] = v; nb[i] = b;
    };
#pragma unroll
    for (int i = 0; i < SPT; i++) { nv[i] = kInfF; nb[i] = 0; tie[i] = false; }
    // Most waves hold ONE type (a wave owns consecutive chunks of the type-sorted net): the code specialised for that kind,
    // two slots at a time -- their reads in flight together, then their arithmetic (all SPT at once: 40 registers more than a
    // 1024-thread workgroup has, 412 bytes of scratch in the frame loop, 17 ms).  The few waves that straddle a type boundary
    // -- they set the pace at the barrier -- take the kind as a run-time scalar and read every source a slot of any kind could
    // have (predecessor ids of a slot without predecessors point at the slot itself), so that their reads are not split by
    // branches either.
    constexpr int G = SPT >= 2 ? 2 : 1;
    if (uniform_kind) {
      switch (ty[0] & kKindMask) {
#define SR_KIND_CASE(K)                                                                                   \
        case K:                                                                                           \
          _Pragma("unroll") for (int h = 0; h < SPT; h += G) {                                            \
            In in[G];                                                                                     \
            _Pragma("unroll") for (int i = h; i < h + G; i++) load_slot(KindC<K>{}, i, in[i - h]);        \
            _Pragma("unroll") for (int i = h; i < h + G; i++) compute_slot(KindC<K>{}, ty[0], i, in[i - h]); \
          }                                                                                               \
          break;
        SR_KIND_CASE(kE0) SR_KIND_CASE(kE0S) SR_KIND_CASE(kE1) SR_KIND_CASE(kE1E) SR_KIND_CASE(kM) SR_KIND_CASE(kME)
#undef SR_KIND_CASE
        default: break;  // kPad
      }
    } else {
#pragma unroll
      for (int h = 0; h < SPT; h += G) {
        In in[G];
#pragma unroll
        for (int i = h; i < h + G; i++) {  // every source a slot of ANY kind could read, unconditionally: no branch, no wait in here
          in[i - h].e = ROWS ? row_l[st[i]] : am[i];
          const uint32_t k_ = ty[i] & kKindMask;
          in[i - h].e0 = ROWS ? row_l[st0[i]] : ((k_ == kE1 || k_ == kE1E) ? am0[i] : am[i]);  // (st0 = st unless position 1)
          in[i - h].c2 = cell(p2[i]);
          in[i - h].c1 = cell(p1[i]);
          in[i - h].c0 = cell(slot_of(i) * 16u);
        }
#pragma unroll
        for (int i = h; i < h + G; i++) {
          switch (ty[i] & kKindMask) {  // wave-uniform; the arithmetic is the specialised one (a run-time kind gets if-converted
                                        // into all kinds' arithmetic plus selects: three times the instructions)
#define SR_KIND_CASE(K) case K: compute_slot(KindC<K>{}, ty[i], i, in[i - h]); break;
            SR_KIND_CASE(kE0) SR_KIND_CASE(kE0S) SR_KIND_CASE(kE1) SR_KIND_CASE(kE1E) SR_KIND_CASE(kM) SR_KIND_CASE(kME)
#undef SR_KIND_CASE
            default: break;  // kPad
          }
        }
      }
    }
    if (tie_any) {  // rare: the boundary source came first where the first minimal word end of the class precedes the slot's word
#pragma unroll
      for (int i = 0; i < SPT; i++) {
        const uint32_t type = ty[i], kind = type & kKindMask;
        const bool b_skip = (kind == kE1 || kind == kE1E) && !(type & kTFirstSil);
        const uint32_t cls = ((type & kTSilWord) ? 0u : 2u) + (b_skip ? 1u : 0u);
        if (tie[i] && ef_cur[cls] < (og[i] >> 16)) nb[i] = bkp_new;  // the in-word candidate had to be strictly better
      }
    }
    if (pad_chunks) {  // padding lanes of a type's last chunk hold nothing
#pragma unroll
      for (int i = 0; i < SPT; i++)
        if (pad_chunks >> i & 1u) nv[i] = og[i] != 0xFFFFFFFFu ? nv[i] : kInfF;
    }
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      const uint32_t kind = ty[i] & kKindMask;
      my_best = dmin(my_best, nv[i]);
      if (kind == kE0S || kind == kE1E || kind == kME) my_we = dmin(my_we, nv[i]);
    }

    // ---- B: block minima through LDS ds_min_f64 cells, fed by one lane per row ------------------------------------
    my_best = row_min_dpp(my_best);
    if (wave_has_we) {  // (wave-uniform; the atomics and their wait are one statement, dpp_util.h)
      my_we = row_min_dpp(my_we);
      if ((lane & 15u) == 0) publish_min2_f64_lds(&c_best[par], my_best, &c_we[par], my_we);
    } else if ((lane & 15u) == 0) {
      publish_min_f64_lds(&c_best[par], my_best);
    }
    __syncthreads();

    // ---- C: prune, publish the word-end minimum --------------------------------------------------------------------
    const double best = c_best[par], we = c_we[par];
    const double limit = best + thr;
    const bool we_alive = !(we > limit) && we != kInfF;
    m_we = we_alive ? we : kInfF;
    prev_alive = we_alive;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      double v = nv[i];
      if (v > limit) v = kInfF;  // :194-196
      nv[i] = v;
      put_cell(slot_of(i) * 16u, v, nb[i]);
    }
    if (wave_has_we && we_alive) {  // wave-uniform
      const double near = m_we + (fabs(m_we) + fabs(wp_word) + fabs(tf) + fabs(ts) + 1.0) * 1e-9;
      bool nr[SPT];
      uint64_t any_near = 0;
#pragma unroll
      for (int i = 0; i < SPT; i++) {
        const uint32_t kind = ty[i] & kKindMask;
        nr[i] = (kind == kE0S || kind == kE1E || kind == kME) && nv[i] <= near;
        any_near |= __ballot(nr[i]);
      }
      if (any_near) {  // the wave that holds the minimum (about one lane of the block)
#pragma unroll
        for (int i = 0; i < SPT; i++) {
          if (nr[i]) {
            const double v = nv[i];
            const uint32_t o = og[i] & 0xFFFFu;
            const bool is_min = v == m_we;
            if (is_min) {  // traceback[t] = the FIRST minimal surviving word end (:199-205): settled by the atomic, written next frame
              // (a lane may hold several of them: it keeps the one with the smallest original index, the only one that can win)
              if (!pend || o < pend_o) { pend_o = o; pend_p = slot_of(i); pend_b = nb[i]; pend_v = v; }
              pend = true;
            }
            // all five atomic minima in one statement (dpp_util.h: no wave-reduction loops, the wait included); 0xFFFFFFFF = nothing
            const uint32_t none = 0xFFFFFFFFu;
            lds_min5_u32(&c_widx[par], is_min ? o : none, ef_nxt,
                         v + 0.0 + tf == m_we + 0.0 + tf ? o : none, v + 0.0 + ts == m_we + 0.0 + ts ? o : none,
                         v + wp_word + tf == m_we + wp_word + tf ? o : none, v + wp_word + ts == m_we + wp_word + ts ? o : none);
          }
        }
      }
    }
    if (ROWS) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next row have landed; the barrier publishes them
    __syncthreads();
  }
  if (T > 0) flush_pending(T);

  // the fast path's premise failed somewhere (a negative or NaN emission cost): hand the utterance to the replay variant
  if (bad) *s_bad = 1;
  __threadfence();
  __syncthreads();
  if (*s_bad) {  // workgroup-uniform
    if (tid == 0) { atomicOr(&a.out_flags[u], kFlagReplay); a.out_count[u] = 0; }
    return;
  }

  // ---- traceback (Recognizer.cpp:222-231; guarded walk: traceback.h) -------------------------------------------
  bool corrupt = false;
  for (uint32_t t = 1 + tid; t <= T; t += NT) {  // winning slot -> word
    const uint32_t sl = __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint16_t w = 0;  // 0xFFFF: no surviving word end -> Book(inf, 0, 0), :118,191
    if (sl != 0xFFFFu) { if (sl < net.n_slots) w = (uint16_t)net.word[sl]; else corrupt = true; }
    a.tb_word[tb0 + t] = w;
  }
  if (corrupt) atomicOr(&a.out_flags[u], kFlagCorrupt);
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

static size_t fast_smem(uint32_t PP, uint32_t ld, bool rows) {
  const size_t base = (fast_smem_base(PP) + 1023u) & ~(size_t)1023u;
  return rows ? base + 2 * (((size_t)ld * 8 + 1023u) & ~(size_t)1023u) : base;
}
static constexpr size_t kLdsPerWorkgroup = 160 * 1024;

hipError_t launch_decode_fast(const DecodeArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const uint32_t P = a.fast.n_slots;
  const dim3 grid(a.n_utts);
#define SR_LAUNCH(NT, SPT)                                                                                              \
  do {                                                                                                                  \
    const bool rows = fast_smem((NT) * (SPT), a.ld, true) <= kLdsPerWorkgroup;                                          \
    const size_t smem = fast_smem((NT) * (SPT), a.ld, rows);                                                            \
    const void* fn = rows ? (const void*)decode_fast_kernel<NT, SPT, true> : (const void*)decode_fast_kernel<NT, SPT, false>; \
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                      \
    if (e != hipSuccess) return e;                                                                                      \
    if (rows) hipLaunchKernelGGL((decode_fast_kernel<NT, SPT, true>), grid, dim3(NT), smem, stream, a);                 \
    else hipLaunchKernelGGL((decode_fast_kernel<NT, SPT, false>), grid, dim3(NT), smem, stream, a);                     \
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
