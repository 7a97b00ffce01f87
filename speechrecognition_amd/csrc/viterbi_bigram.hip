// viterbi_bigram.hip -- bigram-LM beam search over a linear lexicon, one workgroup per utterance.
//
// Follows Teaching::LinearSearch (rwth-asr-0.5/src/Teaching/LinearSearch.cc:211-436,496-515; BookKeeping.cc).
// PARITY UNPINNED: the toolkit does not build here and holds no fixtures; the CPU restatement of the same source in
// the test infrastructure is the specification, tests/test_bigram.py compares against it bit for bit.
//
// The reference keeps hypotheses in lists whose ORDER decides ties and -- through mergeSilenceToBigramNodes' cut of
// the list to its first nWordEnds entries (:378-395) -- even which word ends survive.  The kernel therefore keeps the
// same lists: `L` = the active word hypotheses in activation order (compacted with order-preserving prefix sums, as
// pruneStatesAndFindWordEnds does in place), word ends emitted in that order.  State hypotheses are dense per slot
// (word or silence copy) in LDS: a missing hypothesis is +inf, which no finite candidate loses to, so list
// membership and "score < inf" are the same thing.
//
// Per frame t = 1..T:
//   1 bigramRecombination (:219-244): thread per word w, loop over the ordered word ends (strict <: first wins); the LM
//     table is stored transposed, lmT[h][w], so that the loop reads contiguous rows.  Silence copies take the word
//     end's score.  LM beam (:498-503), entries that fail it are dropped.
//   2 activation (:257-268): newly started words are appended to L in the order of the start list -- words ascending,
//     then silence copies in word-end order (two ordered scans).
//   3 expandHypotheses + addAcousticScores (:270-339): thread per active word, states descending in place; candidates
//     in ascending predecessor order with >= (the later one wins ties); best score by workgroup reduction.
//   4 pruneStatesAndFindWordEnds (:341-376): acoustic beam on score + exit penalty; ordered compaction of L, ordered
//     word-end list.
//   5 mergeSilenceToBigramNodes (:378-395) with its positional quirk, 6 addBookKeepingEntries (:397-418) into an
//     append-only book (the reference's mark-and-sweep only recycles unreachable entries).
// End: traceback from the first best word end (:420-436).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

static constexpr int kBgThreads = 1024;
static constexpr int kBgWaves = kBgThreads / 64;
static constexpr int kBgStage = 256;  // word ends staged per pass of the recombination loop (more for big lexica, see `stage`)
static constexpr float kFltMax = 3.402823466e+38f;
// flags on a word end of the merged list (see step 6): the same word end also sits LATER in the list / sat EARLIER
static constexpr uint32_t kShadowed = 0x80000000u, kRepeat = 0x40000000u, kSlotMask = 0x3FFFFFFFu;

// LDS image without the state hypotheses: entries, staging, scan scratch, the two active lists, the flags
__host__ __device__ constexpr size_t bigram_lds_small(uint32_t n_words) {
  return 2 * (size_t)n_words * 8 + (3 * kBgStage + kBgWaves + 1 + kBgWaves) * 4 + 2 * (size_t)n_words * 2 * 2 + 2 * (size_t)n_words + 64;
}

// register layout: offset of the emission row behind the small image and the LM row bounds
__host__ __device__ constexpr size_t bigram_lds_row_off(uint32_t n_words) {
  return (((bigram_lds_small(n_words) + 15u) & ~(size_t)15u) + 2 * (size_t)n_words * 4 + 1023u) & ~(size_t)1023u;
}

// inclusive prefix sum over the wave with DPP row shifts and broadcasts (six vector instructions; __shfl_up is a
// ds_bpermute round trip per step)
__device__ inline uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1, zero fill
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
  return v;
}

// exclusive prefix of `v` over the workgroup in thread order; total in *total.  `tmp` = kBgWaves + 1 words of LDS.
// ONE barrier: every wave scans the kBgWaves wave totals itself (round 2: a third barrier around a serial loop of thread 0
// over the 16 totals, ~2 000 cycles, five times a frame).
__device__ inline uint32_t wg_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (a scalar: v_readlane below instead of a ds_bpermute round trip)
  const uint32_t inc = wave_incl_scan(v, lane);
  // (no barrier before the write: every call site has a workgroup barrier between the previous scan's reads of `tmp` -- they
  // follow its barrier at once -- and this one)
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  const uint32_t wt = lane < kBgWaves ? tmp[lane] : 0u;
  const uint32_t winc = wave_incl_scan(wt, lane);  // lanes 0 .. kBgWaves-1: inclusive prefix of the wave totals
  *total = (uint32_t)__builtin_amdgcn_readlane((int)winc, kBgWaves - 1);
  const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(winc - wt), wave);  // exclusive prefix of this wave
  return before + inc - v;
}

// minimum over the workgroup (fminf semantics: a NaN loses), returned to every thread; one barrier.  Round 4: the wave and row
// reductions through DPP (ten vector instructions) instead of ten __shfl_xor = ten dependent ds_bpermute round trips -- about
// 1 200 of the 3 400-3 700 cycles each of the three calls per frame cost (tools/bigram_stamps_r4.py, profiles/r4_bigram_steps.txt).
// v_min_f32 directly.  fminf() canonicalises both operands first (a v_max_f32 x, x each: three instructions per minimum, 68 in the
// kernel); the hardware minimum already returns the other operand for a (quiet) NaN, and every operand here is the result of an
// addition or a copy of one, never a signalling NaN.
__device__ inline float fmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_f(float v) {  // lanes without a source keep v
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ inline float wg_min(float v, float* tmp) {
  v = fmin_raw(v, dpp_f<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
  v = fmin_raw(v, dpp_f<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
  v = fmin_raw(v, dpp_f<0x141, 0xF>(v));   // row_half_mirror
  v = fmin_raw(v, dpp_f<0x140, 0xF>(v));   // row_mirror: every lane holds its row's minimum
  v = fmin_raw(v, dpp_f<0x142, 0xA>(v));   // row_bcast15 -> rows 1, 3
  v = fmin_raw(v, dpp_f<0x143, 0xC>(v));   // row_bcast31 -> rows 2, 3: lane 63 holds the wave's minimum
  if ((threadIdx.x & 63) == 63) tmp[threadIdx.x >> 6] = v;  // (as in wg_excl_scan: a barrier lies between two uses of `tmp`)
  __syncthreads();
  float r = tmp[threadIdx.x & (kBgWaves - 1)];  // one read; the 16 partials sit in every row of 16 lanes
  static_assert(kBgWaves == 16, "the cross-wave stage is a reduction over one row of 16 lanes");
  r = fmin_raw(r, dpp_f<0xB1, 0xF>(r));
  r = fmin_raw(r, dpp_f<0x4E, 0xF>(r));
  r = fmin_raw(r, dpp_f<0x141, 0xF>(r));
  r = fmin_raw(r, dpp_f<0x140, 0xF>(r));
  return r;
}

// Two layouts of the state hypotheses (steps 3 and 4):
//   KS == 0  dense in LDS, one thread per POSITION (KP positions per thread): any lexicon whose image fits the LDS;
//   KS >  0  in REGISTERS, one lane per SLOT: lane tid owns the words tid + k * 1024 (k < KS = KW, at most NP states each) and their
//            silence copies (one state: the layout asks for a one-state silence) -- lexica of short words (every BASELINE
//            configuration).  A slot's transitions stay inside the slot, so the expansion is arithmetic on
//            the lane's own registers in descending state order: no LDS image of the states (85 KB at configs[4]), no "all old
//            states read" barrier, no atomics for the survivor flags; the frame's emission costs are staged in LDS by LDS-DMA
//            (one row buffer, fetched while steps 4-6 and 1-2 of the next frame run).  The final state of a surviving word end is
//            published through the entry arrays, which are dead between step 3 and step 5.
// NPM (register layout): 0 = every word has at most three states; else bit k = slot row k (the words k * 1024 .. k * 1024 + 1023) holds a
// word of four states -- only those rows carry the fourth state's registers and arithmetic (round 4; configs[4]'s lexicon has ONE
// four-state word, in row 2: rows 0 and 1 are three-state rows).
template <int KW, int KP, int KS, int NPM>  // KW words per thread in the recombination: W <= KW * kBgThreads
__global__ __launch_bounds__(kBgThreads) void bigram_kernel(BigramArgs a) {
  constexpr bool REGS = KS > 0;
  constexpr int NP = NPM ? 4 : 3;  // states a lane keeps per word at most
  auto np_of = [](int k) constexpr { return ((NPM >> k) & 1) ? 4 : 3; };  // ... and in slot row k
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t W = a.n_words, W2 = 2 * a.n_words, P2 = REGS ? 0u : a.n_positions, sil = a.silence;
  float* st_score = reinterpret_cast<float*>(smem);           // [P2]   (dense layout only)
  uint32_t* st_bp = reinterpret_cast<uint32_t*>(st_score + P2);  // [P2]
  float* en_score = reinterpret_cast<float*>(st_bp + P2);     // [2W] entry (start) hypotheses; scratch in step 5
  uint32_t* en_bp = reinterpret_cast<uint32_t*>(en_score + W2);  // [2W]
  uint32_t* stage = en_bp + W2;                               // [3 * kBgStage]
  uint32_t* scan_tmp = stage + 3 * kBgStage;                  // [kBgWaves + 1]
  float* red_tmp = reinterpret_cast<float*>(scan_tmp + kBgWaves + 1);  // [kBgWaves]
  uint16_t* L[2];
  L[0] = reinterpret_cast<uint16_t*>(red_tmp + kBgWaves);     // [2W] active word slots, activation order
  L[1] = L[0] + W2;
  uint8_t* active = reinterpret_cast<uint8_t*>(L[1] + W2);    // [2W]
  // register layout: the LM row bounds of the skip test (2 x W floats: from global memory they are a dependent round trip in
  // three passes of step 1), then the frame's emission costs (ld doubles), 1 KB aligned
  // -- as bf16, the minimum rounded down and the maximum up: bounds stay bounds, the skip test stays exact (it only drops word ends
  // that cannot win), and the 2 x 2W bytes saved hold the NEW word-end list of step 4 (slots only: score and back pointer are
  // fin_score / fin_bp of the slot), which steps 5 and 6 then read from LDS instead of from the global work space
  uint16_t* lm_lo_lds = reinterpret_cast<uint16_t*>(smem + ((bigram_lds_small(a.n_words) + 15u) & ~(size_t)15u));
  uint16_t* lm_hi_lds = lm_lo_lds + W;
  uint16_t* pm_slot = lm_hi_lds + W;  // [2W]
  uint32_t* n_pairs = reinterpret_cast<uint32_t*>(active + ((W2 + 3u) & ~3u));  // histories whose word AND silence copy end this frame
  unsigned char* row_lds = smem + bigram_lds_row_off(a.n_words);
  auto rowmin = [&](uint32_t h) -> float { if constexpr (REGS) return __uint_as_float((uint32_t)lm_lo_lds[h] << 16); else return a.lm_rowmin[h]; };
  auto rowmax = [&](uint32_t h) -> float { if constexpr (REGS) return __uint_as_float((uint32_t)lm_hi_lds[h] << 16); else return a.lm_rowmax[h]; };
  float* fin_score = en_score;                                // [2W] final-state score / back pointer of the slots whose word end survived
  uint32_t* fin_bp = en_bp;                                   //      (written in step 4, read by its compaction; the entries are consumed by then)

  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x, tid = threadIdx.x;
  const uint64_t f0 = a.frame_off[u], T = a.frame_off[u + 1] - f0;
  const double* dense = a.scores + (f0 - a.frame_base) * a.ld;
  // per-utterance global workspaces
  uint32_t* we_slot[2]; float* we_score[2]; uint32_t* we_bp[2];
  for (int i = 0; i < 2; i++) {
    const uint64_t o = ((uint64_t)blockIdx.x * 2 + i) * W2;
    we_slot[i] = a.we_slot + o; we_score[i] = a.we_score + o; we_bp[i] = a.we_bp + o;
  }
  uint4* book = a.book + a.book_off[u];  // (word, score bits, backpointer, time)
  const uint64_t book_cap = a.book_off[u + 1] - a.book_off[u];
  const float exit_pen[2] = {a.tdp[0][3], a.tdp[1][3]};

  auto map_copy = [&](uint32_t w) { return w == sil ? w : (w >= W ? w - W : w); };
  auto sil_copy = [&](uint32_t w) { return w == sil ? w : w + W; };
  auto ac_word = [&](uint32_t w) { return w < W ? w : sil; };
  auto is_sil = [&](uint32_t w) { return (w == sil || w >= W) ? 1 : 0; };

  for (uint32_t i = tid; i < P2; i += kBgThreads) { st_score[i] = __builtin_inff(); st_bp[i] = 0; }
  for (uint32_t i = tid; i < W2; i += kBgThreads) active[i] = 0;
  // register layout: word w = tid + k * 1024 (slot w) and its silence copy (slot W + w): state count, silence flag, row offsets of the
  // states, hypotheses
  constexpr int KSR = REGS ? KS : 1, NPR = REGS ? NP : 1;
  uint32_t r_n[KSR], r_st[KSR][(NPR + 1) / 2], r_bp[KSR][NPR], rc_bp[KSR];  // r_st: two 16-bit state ids per register
  float r_sc[KSR][NPR], rc_sc[KSR];
  bool r_sil[KSR];
  const uint32_t lane = tid & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // a SCALAR: the piece loop below runs on the scalar unit
  const uint32_t row_bytes = a.ld * 8u;
  const uint32_t sil_st = REGS ? (a.pos_info[a.slot_off[sil]] & 0xFFFFu) * 8u : 0u;  // row offset of the silence state (every copy's state)
  auto issue_row = [&](uint64_t frame /* 1-based */) {  // every wave copies its share of the row's 1 KB pieces (LDS-DMA: no registers)
    const unsigned char* src = reinterpret_cast<const unsigned char*>(dense + (frame - 1) * a.ld) + lane * 16u;
    const uint32_t n_full = row_bytes >> 10;
    uint32_t piece = wave;  // (round 4: with the wave index in a vector register this was an exec-masked vector loop, 18 instructions per piece)
    for (; piece < n_full; piece += kBgWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u),
                                       (__attribute__((address_space(3))) void*)(row_lds + piece * 1024u), 16, 0, 0);
    if (piece == n_full && lane * 16u < (row_bytes & 1023u))  // (a row is a multiple of 64 bytes: the last piece may be short)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u),
                                       (__attribute__((address_space(3))) void*)(row_lds + piece * 1024u), 16, 0, 0);
  };
  if (REGS) {
    for (uint32_t i = tid; i < W; i += kBgThreads) {
      // float -> bf16 by truncation moves a value toward zero: down for x >= 0, up for x < 0; step the other way where asked
      auto to_bf16 = [](float x, bool up) -> uint16_t {
        const uint32_t b = __float_as_uint(x);
        if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)0x7FC0u;  // NaN stays NaN: nothing is skipped against it
        uint32_t t = b >> 16;
        if ((b & 0xFFFFu) && (bool)(b >> 31) != up) t += 1u;  // grow the magnitude: further down for negatives, further up for positives
        return (uint16_t)t;
      };
      lm_lo_lds[i] = to_bf16(a.lm_rowmin[i], false);
      lm_hi_lds[i] = to_bf16(a.lm_rowmax[i], true);
    }
#pragma unroll
    for (int k = 0; k < KSR; k++) {
      const uint32_t w = tid + (uint32_t)k * kBgThreads;
      const bool in = w < W;
      const uint32_t j0 = in ? a.slot_off[w] : 0u;
      r_n[k] = in ? a.slot_off[w + 1] - j0 : 0u;
      r_sil[k] = in && w == sil;
#pragma unroll
      for (int p = 0; p < (NPR + 1) / 2; p++) r_st[k][p] = 0;
#pragma unroll
      for (int p = 0; p < NPR; p++) {
        if ((uint32_t)p < r_n[k]) r_st[k][p / 2] |= (a.pos_info[j0 + p] & 0xFFFFu) << (16 * (p & 1));
        r_sc[k][p] = __builtin_inff();
        r_bp[k][p] = 0;
      }
      rc_sc[k] = __builtin_inff();
      rc_bp[k] = 0;
    }
    if (T > 0) issue_row(1);
  }
  // initialize (:211-216, :397-418 at t = 0): book[0] = sentinel, book[1] = (silence, 0, self, 0); one word end
  uint32_t n_book = 2, n_we = 1, n_L = 0;
  int cur = 0, lcur = 0;  // we_*[cur] = current word ends, L[lcur] = current active list
  if (tid == 0) {
    book[0] = make_uint4(0xFFFFFFFFu, __float_as_uint(kFltMax), 0u, 0u);
    book[1] = make_uint4(sil, __float_as_uint(0.0f), 1u, 0u);
    we_slot[0][0] = sil; we_score[0][0] = 0.0f; we_bp[0][0] = 1u;
  }
  __syncthreads();
  bool overflow = false;
  const __amdgpu_buffer_rsrc_t info_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.pos_info), 0, (int)(P2 * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t slot_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.pos_slot), 0, (int)(P2 * 4u), 0x00020000);
  // The merged word-end list of the previous frame, the thread's own entries [tid * ce, (tid + 1) * ce): in the register layout step 6
  // hands them over in these registers (round 4: the thread that writes an entry is the thread that reads it next frame -- step 6 used
  // to walk the list strided, step 1 read it back from HBM, a dependent round trip of ~2 000 cycles at the top of every frame);
  // the copies in HBM are still written, for the final traceback.
  uint32_t m_raw[KW], m_bp[KW];
  float m_sc[KW];
  for (uint64_t t = 1; t <= T; t++) {
    // ---- 1 bigramRecombination + LM beam ------------------------------------------------------------------------
    for (uint32_t i = W + tid; i < W2; i += kBgThreads) en_score[i] = __builtin_inff();
    if (tid == 0) { en_score[sil] = __builtin_inff(); if (REGS) *n_pairs = 0; }
    float my_score[KW];  // words per thread: W <= KW * kBgThreads
    uint32_t my_bp[KW];
#pragma unroll
    for (int k = 0; k < KW; k++) { my_score[k] = kFltMax; my_bp[k] = 0xFFFFFFFFu; }
    // Word ends that cannot win any word are skipped -- exactly: after the row of word end e every start score is
    // <= score_e + rowmax[h_e], so U = min_e (score_e + rowmax[h_e]) bounds all final start scores; a word end with
    // score_e + rowmin[h_e] > U can never produce "newScore < start[w]" (float addition is monotone, so the bounds hold
    // for the rounded sums).  With a beam of 200 over LM scores spanning ~10 this drops most of the E x W work.
    // The merged word-end list lives in HBM and steps 1 and 2 walk it five times; thread k owns the contiguous entries
    // [k * ce1, (k + 1) * ce1) in every one of those walks (n_we <= W <= KW * 1024: at most KW entries), so it reads them ONCE, into
    // registers: one global round trip (~2 000 cycles) instead of five.
    const uint32_t ce1 = (n_we + kBgThreads - 1) / kBgThreads;
    const uint32_t e1_lo = tid * ce1 < n_we ? tid * ce1 : n_we, e1_hi = (e1_lo + ce1 < n_we) ? e1_lo + ce1 : n_we;
    if (!REGS || t == 1) {  // (register layout: frames 2.. got them from step 6)
#pragma unroll
      for (int i = 0; i < KW; i++) {
        const uint32_t e = e1_lo + (uint32_t)i;
        const bool in = e < e1_hi;
        m_raw[i] = in ? we_slot[cur][e] : 0u;
        m_sc[i] = in ? we_score[cur][e] : 0.0f;
        m_bp[i] = in ? we_bp[cur][e] : 0u;
      }
    }
    float lu = kFltMax;
#pragma unroll
    for (int i = 0; i < KW; i++)
      if (e1_lo + (uint32_t)i < e1_hi) lu = fmin_raw(lu, m_sc[i] + rowmax(map_copy(m_raw[i] & kSlotMask)));
    const float U = wg_min(lu, red_tmp);
    // staging buffer: the dedicated 256 entries, or -- for a big lexicon -- the idle half of the active-list double
    // buffer (it is rewritten from scratch in step 4), up to one entry per thread: fewer passes and barriers
    uint32_t* stg = stage;
    uint32_t stg_cap = kBgStage;
    if ((W2 * 2u) / 12u > (uint32_t)kBgStage) {
      stg = reinterpret_cast<uint32_t*>(L[lcur ^ 1]);
      stg_cap = (W2 * 2u) / 12u < (uint32_t)kBgThreads ? (W2 * 2u) / 12u : (uint32_t)kBgThreads;
    }
    // One pass over the word ends, thread k owning the contiguous entries [k * ce, (k + 1) * ce): the silence-copy transitions, the
    // skip test, and an ordered scan that gives every survivor its place in the staging buffer (round 2 staged chunk after chunk
    // of the list, three barriers per chunk, although on most frames one or two word ends survive the skip test at all).
    uint32_t n_keep = 0;
#pragma unroll
    for (int i = 0; i < KW; i++) {
      if (!(e1_lo + (uint32_t)i < e1_hi)) continue;
      const uint32_t raw = m_raw[i], sl = raw & kSlotMask;
      const float sc_e = m_sc[i];
      // transition into the silence copy of the word that ended (no LM cost); where the merge left the same word end
      // twice in the list, addEntryStateHypothesis (:257-268) keeps the LATER start hypothesis
      if (sl < W && !(raw & kShadowed)) {
        const uint32_t c = sil_copy(sl);
        en_score[c] = sc_e;
        en_bp[c] = m_bp[i];
      }
      n_keep += !(sc_e + rowmin(map_copy(sl)) > U) ? 1u : 0u;
    }
    uint32_t ne_all;
    const uint32_t keep_pos = wg_excl_scan(n_keep, scan_tmp, &ne_all);
    for (uint32_t r0 = 0; r0 < ne_all; r0 += stg_cap) {  // (one round unless more word ends survive than the staging buffer holds)
      if (r0) __syncthreads();  // the previous round's rows have been read
      uint32_t pos = keep_pos;
#pragma unroll
      for (int i = 0; i < KW; i++) {
        if (!(e1_lo + (uint32_t)i < e1_hi)) continue;
        const uint32_t sl = m_raw[i] & kSlotMask, h = map_copy(sl);
        const float sc_e = m_sc[i];
        if (!(sc_e + rowmin(h) > U)) {
          if (pos >= r0 && pos - r0 < stg_cap) {
            stg[3 * (pos - r0)] = h;
            stg[3 * (pos - r0) + 1] = __float_as_uint(sc_e);
            stg[3 * (pos - r0) + 2] = m_bp[i];
          }
          pos++;
        }
      }
      __syncthreads();
      const uint32_t ne = (ne_all - r0 < stg_cap) ? ne_all - r0 : stg_cap;
      // eight word ends at a time (two in the register layout, whose state hypotheses stay live across this loop): their LM rows
      // are loaded first (independent loads in flight together), then compared in list order
      constexpr int KE = REGS ? 2 : 8;
      for (uint32_t e = 0; e < ne; e += KE) {
        float v[KE][KW];
#pragma unroll
        for (int j = 0; j < KE; j++) {
          const uint32_t ej = (e + j < ne) ? e + j : e;
          const float* row = a.lmT + (uint64_t)stg[3 * ej] * W;
#pragma unroll
          for (int k = 0; k < KW; k++) {
            const uint32_t w = tid + k * kBgThreads;
            v[j][k] = row[w < W ? w : 0];
          }
        }
#pragma unroll
        for (int j = 0; j < KE; j++) {
          if (e + j < ne) {  // workgroup-uniform
            const float sc = __uint_as_float(stg[3 * (e + j) + 1]);
            const uint32_t bp = stg[3 * (e + j) + 2];
#pragma unroll
            for (int k = 0; k < KW; k++) {
              const float ns = sc + v[j][k];
              if (ns < my_score[k]) { my_score[k] = ns; my_bp[k] = bp; }
            }
          }
        }
      }
    }
    // (no barrier here: the words' entries written next are read by nobody before the barrier below, and the staging buffer the
    // loop above read is not among them)
    float lmin = kFltMax;
#pragma unroll
    for (int k = 0; k < KW; k++) {
      const uint32_t w = tid + k * kBgThreads;
      if (w < W && w != sil) {
        en_score[w] = my_score[k];
        en_bp[w] = my_bp[k];
        lmin = fmin_raw(lmin, my_score[k]);
      }
    }
    // (no barrier here either: the copies' entries and en_score[sil] read next were written before the scan's barrier above; the
    // words' entries just written are first read in step 2, behind the barrier inside wg_min)
    for (uint32_t i = W + tid; i < W2; i += kBgThreads) lmin = fmin_raw(lmin, en_score[i]);
    if (tid == 0) lmin = fmin_raw(lmin, en_score[sil]);
    const float best_start = wg_min(lmin, red_tmp);
    float lm_thr = a.lm_pruning;
    if (lm_thr < kFltMax) lm_thr += best_start;

    // ---- 2 activation in start-list order: words ascending, then silence copies in word-end order ----------------
    {
      // (a) words: thread k owns the contiguous words [k*cw, (k+1)*cw)
      const uint32_t cw = (W + kBgThreads - 1) / kBgThreads;
      const uint32_t w_lo = tid * cw, w_hi = (w_lo + cw < W) ? w_lo + cw : W;
      uint32_t cnt = 0;
      for (uint32_t w = w_lo; w < w_hi; w++) {
        if (w == sil) continue;  // en_score[sil] is the silence "copy" entered after silence: handled in (b)
        const bool ins = en_score[w] < lm_thr;
        if (!ins) en_score[w] = __builtin_inff();
        else if (!active[w]) cnt++;
      }
      // (b) silence copies, in the order of the word ends that start them.  (a) and (b) touch disjoint slots -- words other than
      // silence there, silence and the copies here -- so both are counted first and ONE scan (16 bits each) places both
      uint32_t cnt_b = 0;
#pragma unroll
      for (int i = 0; i < KW; i++) {  // (the thread's entries of the merged list: in registers since step 1)
        if (!(e1_lo + (uint32_t)i < e1_hi)) continue;
        const uint32_t raw = m_raw[i], sl = raw & kSlotMask;
        if (sl < W && !(raw & kRepeat)) {  // (a repeated word end activates nothing new: its first occurrence did)
          const uint32_t c = sil_copy(sl);
          const bool ins = en_score[c] < lm_thr;
          if (!ins) en_score[c] = __builtin_inff();
          else if (!active[c]) cnt_b++;
        }
      }
      uint32_t total;
      // register layout: this scan's barrier is the last one before step 3 -- the entries are final (the LM beam was applied in the
      // counting passes above), step 3 reads nothing of what the placement below writes (the list and the membership flags are next
      // read behind step 3's own reduction barrier), and the frame's emission row is published here
      if (REGS) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the row have landed
      const uint32_t ex = wg_excl_scan(cnt | (cnt_b << 16), scan_tmp, &total);
      uint32_t pos = n_L + (ex & 0xFFFFu);
      for (uint32_t w = w_lo; w < w_hi; w++)
        if (w != sil && en_score[w] < __builtin_inff() && !active[w]) { L[lcur][pos++] = (uint16_t)w; active[w] = 1; }
      n_L += total & 0xFFFFu;
      pos = n_L + (ex >> 16);
#pragma unroll
      for (int i = 0; i < KW; i++) {
        if (!(e1_lo + (uint32_t)i < e1_hi)) continue;
        const uint32_t raw = m_raw[i], sl = raw & kSlotMask;
        if (sl < W && !(raw & kRepeat)) {
          const uint32_t c = sil_copy(sl);
          if (en_score[c] < __builtin_inff() && !active[c]) { L[lcur][pos++] = (uint16_t)c; active[c] = 1; }
        }
      }
      n_L += total >> 16;
      if (!REGS) __syncthreads();
    }

    // ---- 3 expandHypotheses + addAcousticScores, one thread per POSITION (round 3) --------------------------------------
    // Position j = tid + k * 1024: consecutive lanes hold consecutive positions, i.e. consecutive states of a word -- the LDS
    // accesses are conflict-free and the emission gather of a wave is a few contiguous cache lines.  (Rounds 1-2 walked the
    // active list, one thread per slot: neighbouring threads were 15-16 positions apart = a 16-way bank conflict on every
    // state access, and every emission gather touched 64 cache lines; the step took 34 000 of a frame's 104 000 cycles.)
    // A slot that is not active holds +inf everywhere and has no entry hypothesis, so updating it is a no-op; the new state s
    // needs the OLD states s-2, s-1, s: all reads, a barrier, then the writes.  Candidates in the order the reference creates
    // them (ascending predecessor state, entry first); >= lets the later one win.
    float lbest = kFltMax;
    constexpr int KPD = REGS ? 1 : KP;
    float nsc[KPD];
    uint32_t nbp[KPD];
    if constexpr (REGS) {
      // ---- register layout: the lane's words, states descending in place (new state s needs the old s, s - 1, s - 2), then their
      // silence copies (one state: entry or loop) ------------------------------------------------------------------------------
      const float inf = __builtin_inff();
      const float pem_sil = (float)*reinterpret_cast<const double*>(row_lds + sil_st);
#pragma unroll
      for (int k = 0; k < KS; k++) {
        const uint32_t w = tid + (uint32_t)k * kBgThreads;
        const bool sil_ = r_sil[k];
        // the silence word has ONE state in this layout (bigram_register_layout): of its penalties only the loop applies -- forward
        // and skip into states it does not have are computed and discarded (`in` below) -- so the words' are scalars for every lane
        const float t0 = sil_ ? a.tdp[1][0] : a.tdp[0][0], t1 = a.tdp[0][1], t2 = a.tdp[0][2];
        const float ent = w < W ? en_score[w] : inf;
        const uint32_t ebp = w < W ? en_bp[w] : 0u;
        float pem[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
          if (p < np_of(k)) pem[p] = (float)*reinterpret_cast<const double*>(row_lds + ((r_st[k][p / 2] >> (16 * (p & 1))) & 0xFFFFu) * 8u);
#pragma unroll
        for (int p = NP - 1; p >= 0; p--) {
          if (p >= np_of(k)) continue;  // (compile time: this row has no fourth state)
          const bool in = (uint32_t)p < r_n[k];
          float best = inf;
          uint32_t bb = 0;
          // candidates in ascending predecessor order, >= lets the later one win: the virtual entry state 0 (free into state 1,
          // skip penalty into state 2), then s - 2, s - 1, s
          if (p == 0) { if (ent < inf) { best = ent; bb = ebp; } }
          if (p == 1) { if (ent < inf) { best = ent + t2; bb = ebp; } }
          if (p >= 2) { const float o = r_sc[k][p - 2], c = o + t2; if (o < inf && !(best < c)) { best = c; bb = r_bp[k][p - 2]; } }
          if (p >= 1) { const float o = r_sc[k][p - 1], c = o + t1; if (o < inf && !(best < c)) { best = c; bb = r_bp[k][p - 1]; } }
          { const float o = r_sc[k][p], c = o + t0; if (o < inf && !(best < c)) { best = c; bb = r_bp[k][p]; } }
          if (!in) best = inf;
          if (best < inf) {
            best += pem[p];
            lbest = fmin_raw(lbest, best);
          }
          r_sc[k][p] = best;
          r_bp[k][p] = best < inf ? bb : 0u;
        }
        // the copy: slot W + w (the silence word itself has none)
        const bool has_copy = w < W && w != sil;
        const float cent = has_copy ? en_score[W + w] : inf;
        const uint32_t cebp = has_copy ? en_bp[W + w] : 0u;
        float best = inf;
        uint32_t bb = 0;
        if (cent < inf) { best = cent; bb = cebp; }
        { const float o = rc_sc[k], c = o + a.tdp[1][0]; if (o < inf && !(best < c)) { best = c; bb = rc_bp[k]; } }
        if (best < inf) {
          best += pem_sil;
          lbest = fmin_raw(lbest, best);
        }
        rc_sc[k] = best;
        rc_bp[k] = best < inf ? bb : 0u;
      }
    } else {
    const double* row = dense + (t - 1) * a.ld;
    // kCh positions at a time: emission state | flags, slot (frame-invariant, but held in registers across the frame loop they
    // spill: coalesced reloads are cheaper) and emission cost, loads in flight together, then the four updates.
    constexpr int kCh = 4;
    static_assert(KP % kCh == 0, "positions per thread come in chunks of four");
#pragma unroll
    for (int k0 = 0; k0 < KP; k0 += kCh) {
    __builtin_amdgcn_sched_barrier(0);  // (chunks interleaved by the scheduler need more registers than there are)
    uint32_t pinfo[kCh], pslot[kCh];
    float pem[kCh];
#pragma unroll
    for (int q = 0; q < kCh; q++) {
      const uint32_t j = tid + (uint32_t)(k0 + q) * kBgThreads;
      // (buffer loads: one VGPR offset for all positions, the k * 4096 bytes as a scalar offset -- as plain pointers the compiler
      // hoists KP 64-bit addresses per table out of the frame loop and spills them; out of range reads return 0)
      pinfo[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(info_rsrc, tid * 4u, (uint32_t)(k0 + q) * (kBgThreads * 4u), 0);
      pslot[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(slot_rsrc, tid * 4u, (uint32_t)(k0 + q) * (kBgThreads * 4u), 0);
      (void)j;
    }
#pragma unroll
    for (int q = 0; q < kCh; q++) pem[q] = (float)row[pinfo[q] & 0xFFFFu];
    // Four positions at a time, branch-free: (a) every candidate's old score is read unconditionally (an address that does not
    // apply is clamped to the position itself, its value replaced by +inf), (b) the decisions, pure arithmetic, (c) the winners'
    // back pointers -- two LDS round trips per four positions instead of five per position behind divergent branches.
    constexpr int kL = 4;
    static_assert(kCh % kL == 0, "");
#pragma unroll
    for (int q0 = 0; q0 < kCh; q0 += kL) {
      float c_ent[kL], c_o3[kL], c_o2[kL], c_o1[kL];
#pragma unroll
      for (int r = 0; r < kL; r++) {
        const uint32_t j = tid + (uint32_t)(k0 + q0 + r) * kBgThreads, jj = j < P2 ? j : 0u;
        const uint32_t fl = pinfo[q0 + r] >> 16;
        c_ent[r] = en_score[pslot[q0 + r]];
        c_o3[r] = st_score[(fl & 3u) ? jj : jj - 2];
        c_o2[r] = st_score[(fl & 1u) ? jj : jj - 1];
        c_o1[r] = st_score[jj];
      }
      uint32_t srcs[kL];  // where the winner's back pointer lives: 0xFFFFFFFF none, 0x80000000 | slot = entry, else position
#pragma unroll
      for (int r = 0; r < kL; r++) {
        const int q = q0 + r, k = k0 + q;
        const uint32_t j = tid + (uint32_t)k * kBgThreads;
        const uint32_t fl = pinfo[q] >> 16, sl = pslot[q];
        const bool sil_ = (fl >> 3) & 1u, in = j < P2;
        // the dense layout is where a silence word of SEVERAL states lands (bigram_register_layout requires one): inside silence
        // and its copies all three silence penalties apply, tdp[isSilence][s' - s] (LinearSearch.cc:296-326).  Only the register
        // layout above may treat forward and skip as the words' scalars.
        const float t0 = sil_ ? a.tdp[1][0] : a.tdp[0][0], t1 = sil_ ? a.tdp[1][1] : a.tdp[0][1], t2 = sil_ ? a.tdp[1][2] : a.tdp[0][2];
        const float inf = __builtin_inff();
        // states 1 and 2 are reachable from the virtual entry state 0: free to state 1, skip penalty to state 2
        const float ent = (in && (fl & 3u)) ? c_ent[r] : inf;
        const float o3 = (in && !(fl & 3u)) ? c_o3[r] : inf;  // s >= 3: skip from s - 2
        const float o2 = (in && !(fl & 1u)) ? c_o2[r] : inf;  // s >= 2: forward from s - 1
        const float o1 = in ? c_o1[r] : inf;                  // loop
        float best = inf;
        uint32_t src = 0xFFFFFFFFu;
        if (ent < inf) { best = (fl & 1u) ? ent : ent + t2; src = 0x80000000u | sl; }
        { const float c = o3 + t2; if (o3 < inf && !(best < c)) { best = c; src = j - 2; } }
        { const float c = o2 + t1; if (o2 < inf && !(best < c)) { best = c; src = j - 1; } }
        { const float c = o1 + t0; if (o1 < inf && !(best < c)) { best = c; src = j; } }
        if (best < inf) {
          best += pem[q];
          lbest = fmin_raw(lbest, best);
        }
        nsc[k] = best;
        srcs[r] = src;
      }
#pragma unroll
      for (int r = 0; r < kL; r++) {
        const uint32_t src = srcs[r];
        // en_bp follows st_bp's neighbour arrays in LDS: one read through a common base (st_bp) with a signed word offset
        const uint32_t idx = src == 0xFFFFFFFFu ? 0u : (src & 0x80000000u) ? (uint32_t)(en_bp - st_bp) + (src & 0x7FFFFFFFu) : src;
        const uint32_t v = st_bp[idx];
        nbp[k0 + q0 + r] = src == 0xFFFFFFFFu ? 0u : v;
      }
    }
    }
    __syncthreads();  // every old state has been read
#pragma unroll
    for (int k = 0; k < KP; k++) {
      const uint32_t j = tid + (uint32_t)k * kBgThreads;
      if (j < P2) { st_score[j] = nsc[k]; st_bp[j] = nbp[k]; }
    }
    for (uint32_t i = tid; i < W2; i += kBgThreads) active[i] &= 1u;  // bit 0 = on the active list; bits 1, 2 = this frame's survivors
    }
    const float best_score = wg_min(lbest, red_tmp);
    float ac_thr = a.ac_pruning;
    if (ac_thr < kFltMax) ac_thr += best_score;
    if (REGS && t < T) issue_row(t + 1);  // every wave is past its reads of this frame's row (the barrier inside wg_min)

    // ---- 4 pruneStatesAndFindWordEnds: the acoustic beam per position, then the ordered compaction of the active list ---
    if constexpr (REGS) {
#pragma unroll
      for (int k = 0; k < KS; k++) {
        const uint32_t w = tid + (uint32_t)k * kBgThreads;
        const float pen = r_sil[k] ? a.tdp[1][3] : a.tdp[0][3];
        uint32_t flags = 0;
        float f_sc = 0.0f;
        uint32_t f_bp = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) {
          if (p >= np_of(k)) continue;  // (compile time)
          const float v = r_sc[k][p];
          if (v < __builtin_inff()) {
            if (v + pen < ac_thr) {
              flags |= 2u;
              if ((uint32_t)p + 1u == r_n[k]) { flags |= 4u; f_sc = v; f_bp = r_bp[k][p]; }  // its final state too
            } else {
              r_sc[k][p] = __builtin_inff();
            }
          }
        }
        if (w < W) {  // bit 0 = on the active list; bits 1, 2 = this frame's survivors
          active[w] = (uint8_t)((active[w] & 1u) | flags);
          if (flags & 4u) { fin_score[w] = f_sc; fin_bp[w] = f_bp; }
          uint32_t cflags = 0;
          if (rc_sc[k] < __builtin_inff()) {
            if (rc_sc[k] + a.tdp[1][3] < ac_thr) cflags = 6u;  // (one state: alive = its final state alive)
            else rc_sc[k] = __builtin_inff();
          }
          active[W + w] = (uint8_t)((active[W + w] & 1u) | cflags);
          if (cflags) { fin_score[W + w] = rc_sc[k]; fin_bp[W + w] = rc_bp[k]; }
          if ((flags & 4u) && cflags) atomicAdd(n_pairs, 1u);  // both word ends of this history survive: the merge keeps one entry for them
        }
      }
    } else {
      uint32_t* active_w = reinterpret_cast<uint32_t*>(active);  // (byte flags, OR-ed through their 32-bit word)
#pragma unroll
      for (int k = 0; k < KP; k++) {
        const uint32_t j = tid + (uint32_t)k * kBgThreads;
        if (j < P2 && nsc[k] < __builtin_inff()) {
          const uint32_t fl = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(info_rsrc, tid * 4u, (uint32_t)k * (kBgThreads * 4u), 0) >> 16;
          const uint32_t sl = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(slot_rsrc, tid * 4u, (uint32_t)k * (kBgThreads * 4u), 0);
          const float tv = nsc[k] + exit_pen[(fl >> 3) & 1];  // (nsc: what this thread stored at st_score[j] above)
          if (tv < ac_thr) atomicOr(&active_w[sl >> 2], ((fl & 4u) ? 6u : 2u) << (8u * (sl & 3u)));  // alive; its final state too
          else st_score[j] = __builtin_inff();
        }
      }
    }
    __syncthreads();
    const uint32_t cl = (n_L + kBgThreads - 1) / kBgThreads;
    const uint32_t i_lo = tid * cl, i_hi = (i_lo + cl < n_L) ? i_lo + cl : n_L;
    uint32_t n_alive = 0, n_ends = 0;
    for (uint32_t i = i_lo; i < i_hi; i++) {
      const uint32_t fa = active[L[lcur][i]];
      n_alive += (fa >> 1) & 1u;
      n_ends += (fa >> 2) & 1u;
    }
    // one scan for both counts (16 bits each: at most 2W = 16 384 slots)
    uint32_t tot_both;
    const uint32_t p_both = wg_excl_scan(n_alive | (n_ends << 16), scan_tmp, &tot_both);
    const uint32_t tot_alive = tot_both & 0xFFFFu, tot_ends = tot_both >> 16;
    uint32_t pa = p_both & 0xFFFFu, pe = p_both >> 16;
    const int nxt = cur ^ 1;
    for (uint32_t i = i_lo; i < i_hi; i++) {
      const uint32_t sl = L[lcur][i];
      const uint32_t fa = active[sl];
      if (!(fa & 2u)) { active[sl] = 0; continue; }
      active[sl] = 1;
      L[lcur ^ 1][pa++] = (uint16_t)sl;
      if (fa & 4u) {  // the final state survived: a word end, carrying the exit-penalised score
        if (REGS) {
          pm_slot[pe] = (uint16_t)sl;  // (score = fin_score[sl] + exit penalty, back pointer = fin_bp[sl]: steps 5 and 6 look them up)
        } else {
          const uint32_t last = a.slot_off[sl + 1] - 1;
          we_slot[nxt][pe] = sl;
          we_score[nxt][pe] = st_score[last] + exit_pen[is_sil(sl)];
          we_bp[nxt][pe] = st_bp[last];
        }
        pe++;
      }
    }
    lcur ^= 1;
    n_L = tot_alive;
    __syncthreads();  // the word-end list in global memory is complete (same workgroup: visible after the barrier)

    // ---- 5 mergeSilenceToBigramNodes.  first[h] / last[h] = first / last index of a word end with history h; the
    // reference writes the better of the (at most two) into the FIRST index and then keeps list[0 .. #histories) --------
    uint32_t n_hist;
    if constexpr (REGS) {
      // A history h has at most two word ends: slot h (the word) and slot W + h (its silence copy).  pos[slot] = the slot's index in
      // the new list (0xFFFF: not a word end this frame), kept in the idle half of the active-list double buffer: first and last
      // are the smaller and the larger of the two positions -- no atomics, and every look-up of steps 5 and 6 is an LDS read.
      // The buffer is not cleared: whatever a slot's cell holds (an older frame's position, a slot number of the old active list),
      // it is this frame's position of slot s if and only if it is below the list's length and the list holds s there.
      uint16_t* pos = L[lcur ^ 1];
      for (uint32_t e = tid; e < tot_ends; e += kBgThreads) pos[pm_slot[e]] = (uint16_t)e;
      __syncthreads();
      auto pos_of = [&](uint32_t s_) -> uint32_t { const uint32_t g = pos[s_]; return (g < tot_ends && pm_slot[g] == s_) ? g : 0xFFFFu; };
      auto first_last = [&](uint32_t h, uint32_t* fi, uint32_t* la) {
        const uint32_t p1 = pos_of(h), p2 = h != sil ? pos_of(W + h) : 0xFFFFu;
        *fi = p1 < p2 ? p1 : p2;
        *la = p1 == 0xFFFFu ? p2 : p2 == 0xFFFFu ? p1 : (p1 > p2 ? p1 : p2);
      };
      auto we_sc = [&](uint32_t e) -> float { const uint32_t sl = pm_slot[e]; return fin_score[sl] + exit_pen[is_sil(sl)]; };
      n_hist = tot_ends - *n_pairs;  // #histories = #word ends - #histories with two of them (counted by their owners in step 4)
      // ---- 6 addBookKeepingEntries for the kept word ends e = 0 .. n_hist-1, in list order ---------------------------
      if ((uint64_t)n_book + n_hist > book_cap) { overflow = true; break; }  // workgroup-uniform
      // thread k owns the contiguous entries [k * ce6, (k + 1) * ce6): the ones step 1 of the next frame gives it (n_hist = the next n_we)
      const uint32_t ce6 = (n_hist + kBgThreads - 1) / kBgThreads;
      const uint32_t e6_lo = tid * ce6 < n_hist ? tid * ce6 : n_hist, e6_hi = (e6_lo + ce6 < n_hist) ? e6_lo + ce6 : n_hist;
#pragma unroll
      for (int i = 0; i < KW; i++) {
        const uint32_t e = e6_lo + (uint32_t)i;
        if (!(e < e6_hi)) { m_raw[i] = 0u; m_sc[i] = 0.0f; m_bp[i] = 0u; continue; }
        uint32_t fi, la;
        first_last(map_copy(pm_slot[e]), &fi, &la);
        uint32_t src = e, mark = 0;
        if (fi == e) {
          if (la != e && we_sc(la) <= we_sc(e)) {  // <=: the later one wins a tie
            src = la;
            if (la < n_hist) mark = kShadowed;  // ... and stays in the list at its own index too
          }
        } else {
          if (we_sc(e) <= we_sc(fi)) mark = kRepeat;  // (fi < e < n_hist) did index fi take over this entry?
        }
        const uint32_t sl = pm_slot[src];
        const float sc = fin_score[sl] + exit_pen[is_sil(sl)];
        uint32_t bp = fin_bp[sl];
        if (sl == sil) {  // avoid chains of silence (:410-415); entries store acoustic words, copies count as silence
          const uint4 prev = book[bp];
          if (prev.x == sil) bp = prev.z;
        }
        const uint32_t nb = n_book + e;
        book[nb] = make_uint4(ac_word(sl), __float_as_uint(sc), bp, (uint32_t)t);
        we_slot[cur][e] = sl | mark; we_score[cur][e] = sc; we_bp[cur][e] = nb;
        m_raw[i] = sl | mark; m_sc[i] = sc; m_bp[i] = nb;
      }
    } else {
      uint32_t* first = reinterpret_cast<uint32_t*>(en_score);  // [W] (entries are consumed; rebuilt next frame)
      uint32_t* last = en_bp;                                   // [W]
      for (uint32_t h = tid; h < W; h += kBgThreads) { first[h] = 0xFFFFFFFFu; last[h] = 0u; }
      __syncthreads();
      for (uint32_t e = tid; e < tot_ends; e += kBgThreads) {
        const uint32_t h = map_copy(we_slot[nxt][e]);
        atomicMin(&first[h], e);
        atomicMax(&last[h], e);
      }
      __syncthreads();
      uint32_t nh = 0;
      for (uint32_t e = tid; e < tot_ends; e += kBgThreads)  // (the second walk hits L1: caching the entries in registers was slower, 35.5 vs 34.4 ms)
        if (first[map_copy(we_slot[nxt][e])] == e) nh++;
      (void)wg_excl_scan(nh, scan_tmp, &n_hist);
      // ---- 6 addBookKeepingEntries for the kept word ends e = 0 .. n_hist-1, in list order -----------------------------
      if ((uint64_t)n_book + n_hist > book_cap) { overflow = true; break; }  // workgroup-uniform
      for (uint32_t e = tid; e < n_hist; e += kBgThreads) {
        const uint32_t h = map_copy(we_slot[nxt][e]);
        uint32_t src = e, mark = 0;
        if (first[h] == e) {
          const uint32_t j = last[h];
          if (j != e && we_score[nxt][j] <= we_score[nxt][e]) {  // <=: the later one wins a tie
            src = j;
            if (j < n_hist) mark = kShadowed;  // ... and stays in the list at its own index too
          }
        } else {
          const uint32_t i = first[h];  // (i < e < n_hist) did index i take over this entry?
          if (we_score[nxt][e] <= we_score[nxt][i]) mark = kRepeat;
        }
        const uint32_t sl = we_slot[nxt][src];
        const float sc = we_score[nxt][src];
        uint32_t bp = we_bp[nxt][src];
        if (sl == sil) {  // avoid chains of silence (:410-415); entries store acoustic words, copies count as silence
          const uint4 prev = book[bp];
          if (prev.x == sil) bp = prev.z;
        }
        const uint32_t nb = n_book + e;
        book[nb] = make_uint4(ac_word(sl), __float_as_uint(sc), bp, (uint32_t)t);
        // the merged list for the next frame goes to the other buffer (reads above are from `nxt`, writes to `cur`)
        we_slot[cur][e] = sl | mark; we_score[cur][e] = sc; we_bp[cur][e] = nb;
      }
    }
    n_book += n_hist;
    n_we = n_hist;
    __syncthreads();
    // we_*[cur] now holds the new word ends: `cur` stays
  }

  // ---- traceback (:420-436): first minimum in list order --------------------------------------------------------
  if (tid == 0) {
    uint32_t n_out = 0;
    if (overflow) {
      a.out_flags[u] = 1;
    } else {
      a.out_flags[u] = 0;
      if (n_we > 0) {
        uint32_t bi = 0;
        float bs = we_score[cur][0];
        for (uint32_t i = 1; i < n_we; i++) { const float s = we_score[cur][i]; if (s < bs) { bs = s; bi = i; } }
        const uint32_t bp0 = we_bp[cur][bi];
        uint32_t len = 0;
        for (uint32_t b = bp0; book[b].w > 0; b = book[b].z) len++;
        n_out = len;
        const uint64_t o = f0 + u;  // T_u + 1 output slots per utterance
        for (uint32_t b = bp0; book[b].w > 0; b = book[b].z) {
          len--;
          const uint4 e = book[b];
          a.out_word[o + len] = e.x; a.out_score[o + len] = __uint_as_float(e.y); a.out_time[o + len] = e.w;
        }
      }
    }
    a.out_count[u] = n_out;
  }
}

size_t bigram_lds_bytes(uint32_t n_words, uint32_t n_positions) { return (size_t)n_positions * 8 + bigram_lds_small(n_words); }
uint32_t bigram_max_words() { return 8 * kBgThreads; }

static size_t bigram_lds_regs(uint32_t n_words, uint32_t ld) { return bigram_lds_row_off(n_words) + (((size_t)ld * 8 + 1023u) & ~(size_t)1023u); }
bool bigram_register_layout(const BigramArgs& a) {
  return !a.dense_states && a.max_slot_states <= 4 && a.silence_states == 1 && a.n_words <= 3 * kBgThreads && bigram_lds_regs(a.n_words, a.ld) <= 160 * 1024;
}

hipError_t launch_bigram(const BigramArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const bool regs = bigram_register_layout(a);
  const size_t smem = regs ? bigram_lds_regs(a.n_words, a.ld) : (bigram_lds_bytes(a.n_words, a.n_positions) + 15) & ~(size_t)15;
  auto go = [&](auto kernel) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(a.n_utts), dim3(kBgThreads), smem, stream, a);
    return hipGetLastError();
  };
  const uint32_t kw = (a.n_words + kBgThreads - 1) / kBgThreads, kp = (a.n_positions + kBgThreads - 1) / kBgThreads;
  if (regs) {  // lane tid: words tid + k * 1024 and their silence copies; three states per word, four in the slot rows of row4_mask
    const uint32_t mask = a.max_slot_states <= 3 ? 0u : (a.row4_mask ? a.row4_mask : (1u << kw) - 1u);
    switch (kw <= 1 ? 1 : kw <= 2 ? 2 : 3) {
      case 1:
        if (mask == 0) return go(bigram_kernel<1, 4, 1, 0>);
        return go(bigram_kernel<1, 4, 1, 1>);
      case 2:
        switch (mask & 3u) {
          case 0: return go(bigram_kernel<2, 4, 2, 0>);
          case 1: return go(bigram_kernel<2, 4, 2, 1>);
          case 2: return go(bigram_kernel<2, 4, 2, 2>);
          default: return go(bigram_kernel<2, 4, 2, 3>);
        }
      default:
        switch (mask & 7u) {
          case 0: return go(bigram_kernel<3, 4, 3, 0>);
          case 1: return go(bigram_kernel<3, 4, 3, 1>);
          case 2: return go(bigram_kernel<3, 4, 3, 2>);
          case 3: return go(bigram_kernel<3, 4, 3, 3>);
          case 4: return go(bigram_kernel<3, 4, 3, 4>);
          case 5: return go(bigram_kernel<3, 4, 3, 5>);
          case 6: return go(bigram_kernel<3, 4, 3, 6>);
          default: return go(bigram_kernel<3, 4, 3, 7>);
        }
    }
  }
#define SR_BG(KWv, KPv) return go(bigram_kernel<KWv, KPv, 0, 0>)
#define SR_BG_KP(KWv) do { if (kp <= 4) SR_BG(KWv, 4); if (kp <= 12) SR_BG(KWv, 12); if (kp <= 20) SR_BG(KWv, 20); return hipErrorInvalidValue; } while (0)
  if (kw <= 1) SR_BG_KP(1);
  if (kw <= 2) SR_BG_KP(2);
  if (kw <= 4) SR_BG_KP(4);
  SR_BG_KP(8);
#undef SR_BG_KP
#undef SR_BG
}

}  // namespace srgpu
