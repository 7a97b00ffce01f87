// viterbi_words.hip -- the beam Viterbi decoder (Recognizer::recognizeSequence_pruned, sietill/Recognizer.cpp:103-232) for
// lexica of SHORT words (every word at most four positions: the synthetic configurations of SURVEY 8d are silence + W words of
// three states).  One workgroup per utterance, one LANE per word (NW words per lane: the host picks NW so that a workgroup has
// at most 8 waves where it can -- two workgroups then share a CU).
//
// In a linear whole-word lexicon every in-word transition stays inside its word (loop, forward, skip: Recognizer.cpp:160-186),
// and the word-boundary transition reaches a word only through the minimum over the surviving word ends (viterbi_decode.hip, the
// exactness argument; viterbi_fast.hip, the collapsed boundary candidate).  A lane that owns a whole word therefore keeps the
// word's hypotheses -- score and back pointer per position -- in REGISTERS over the frame loop: what the slot-per-lane kernel
// (viterbi_fast.hip) moves through LDS per frame, 16 bytes per hypothesis read up to three times and written once, and the
// barrier that separates those reads from the writes, do not exist here.  Per frame:
//
//   A  every lane: emission costs of its positions from the score row in LDS (staged by LDS-DMA, issued two frames ahead right behind the barrier), the new
//      hypotheses from the old ones in registers, in the reference's source order (skip, forward, loop ascending by source
//      index; a later candidate must be strictly better), the boundary candidate (m_we + word penalty) + tdp + position 0's
//      emission for positions 0 and 1 (Recognizer.cpp:133-157, :148-151 for the emission quirk);
//   B  block minimum and word-end minimum: DPP row reduction, four LDS ds_min_f64 per wave;            -- the ONE barrier --
//   C  prune against best + am_threshold (:194-196), the word-end bookkeeping: traceback[t] = the FIRST minimal surviving
//      word end (:199-205, LDS atomic min on the original hypothesis index), first word-end index per boundary class for the
//      tie order of the next frame.
//
// Everything that crosses the barrier sits in LDS cells that rotate over THREE frames (written in frame t, read in t or t + 1,
// reset one barrier before their next use), which is what lets a frame do with one barrier.
//
// Same premise as the slot kernel: every emission cost of the utterance is >= 0 (the reference's pre-AM early-out,
// Recognizer.cpp:143,173, is inert then); an utterance that breaks it is flagged kFlagReplay and redone by
// decode_kernel<.., REPLAY = true>.  Results are bit-identical to the slot kernel, the general kernel and the oracle; tie
// order is the reference's hypothesis index word * max_pos + pos, which orders like word_off[word] + pos.
//
// NEG = true (round 5): the same kernel WITHOUT that premise, for models whose emission costs can be negative (the host decides from
// the model: some density has norm - log weight < 0).  The early-out compares a candidate BEFORE its emission cost with the target's
// value AFTER it: with a negative cost a candidate that would have won can be skipped, and what a slot ends up with depends on the
// order of its candidates.  Three things replace the fast formulas, frame by frame and only where needed:
//   * positions >= 2 make their three offers (skip, forward, loop) through Merge::offer, the reference's own test, always;
//   * positions 0 and 1 (and a one-position word's dead position-1 slot) keep the collapsed boundary candidate while the word's
//     entry emission costs are >= 0 (exact then, see above);
//   * where an entry emission cost is negative or not a number, the lane replays the boundary loop (Recognizer.cpp:133-158) for its
//     word against the surviving word ends of the previous frame in hypothesis order: word ends of the words before it, its own
//     in-word sources, the remaining word ends.  The word-end scores sit in a dense per-word LDS array (written unpruned in phase A,
//     double buffered by frame parity, the previous frame's prune limit applied while reading); a wave scans it 64 words per read,
//     ballots the live ones and visits only those -- O(W / 64 + live word ends) per affected wave and frame instead of the general
//     kernel's O(W) per lane on top of a five times heavier frame (220 ms against 2 ms on configs[2]'s lexicon with tight variances,
//     profiles/r5_cliffs.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dpp_util.h"
#include "kernels.h"
#include "traceback.h"

namespace srgpu {

// word info: bits 0-2 number of positions (0 = padding lane), bit 3 silence word, bit 4 first state is the silence state,
// bits 8-11 position p's state is the silence state
static constexpr uint32_t kWSilWord = 8u, kWFirstSil = 16u, kWSilStates = 0xF00u;

static constexpr uint32_t kWordsCellBytes = 1024;  // minima and first-index cells; the row buffers follow, 1 KB aligned

// Lanes take their words from an ORDER the host chose (srgpu_api.cpp: sr_lexicon_create).  A (wave, k) GROUP of 64 word slots
// holds words of one kind -- the host pads every kind to whole groups -- and runs that kind's code:
//   plain    exactly L positions, not the silence word, no silence state: every transition penalty a scalar, the role of
//            every position known at compile time -- a third of the general path's instructions.  In SURVEY 8d's lexica
//            every word but silence (and cfg5's one four-state word) is plain;
//   single   one-position words (silence): the boundary candidate and the dead position-1 hypothesis, nothing else;
//   general  anything of up to four positions, all decisions per lane (template GEN: compiled in only for lexica that have
//            such words -- its registers would set the kernel's budget); the host gives a general group a wave of its own;
//   skip     padding only.
// A frame lasts as long as its slowest wave (the barrier), so what counts is the instruction count of the heaviest wave.
// Padding lanes of a plain or single group read a +inf emission cost that sits behind the row in LDS: every candidate of theirs
// is +inf without a select.
enum : uint32_t { kGSkip = 0, kGPlain = 1, kGSingle = 2, kGGeneral = 3 };

// one target hypothesis being built, candidates offered in the reference's source order (Recognizer.cpp:143-157 / :173-186): the
// early-out on the candidate BEFORE its emission cost, then strict improvement
struct WMerge {
  double score;
  uint32_t bkp;
  __device__ WMerge() : score(kInfF), bkp(0) {}
  __device__ void offer(double pre_am, double am, uint32_t cand_bkp) {
    const double n = pre_am + am;
    const bool take = !(pre_am > score) && (score > n);
    score = take ? n : score;
    bkp = take ? cand_bkp : bkp;
  }
};
__device__ inline double readlane_d(double v, uint32_t l) {  // l wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (int)l), __builtin_amdgcn_readlane(__double2loint(v), (int)l));
}

// MAXT: the launch bound.  1024 (128 registers: two workgroups of <= 8 waves per CU) for the fast variant; the NEG variant's word-end
// arrays leave room for one workgroup per CU only, so with <= 512 threads it may use 256 registers (no spills).
template <int NW, int L, bool GEN, bool NEG = false, int MAXT = 1024>
__global__ __launch_bounds__(MAXT) void decode_words_kernel(DecodeArgs a) {
  constexpr int NPA = GEN ? 4 : L;  // positions a lane keeps per word
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* c_best = reinterpret_cast<double*>(smem);                 // [3] block minimum of the frame's new scores, by frame mod 3
  double* c_we = c_best + 3;                                        // [3] minimum over the word-end hypotheses
  uint32_t* c_widx = reinterpret_cast<uint32_t*>(c_we + 3);         // [3] first original index among the minimal word ends
  uint32_t* e_first = c_widx + 3;                                   // [3][4] first word-end original index per boundary class
  uint32_t* s_bad = e_first + 12;                                   // [1]
  unsigned char* rows_lds = smem + kWordsCellBytes;
  const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, n_waves = nt >> 6;
  // the wave's index as a SCALAR: the compiler cannot know that threadIdx.x >> 6 is wave-uniform, and with it in a vector register
  // the piece loop of issue_row below became an exec-masked vector loop with a v_readfirstlane per piece -- 18 instructions per
  // 1 KB piece, 15 % of a frame at configs[2] (tools/words_stamps_r4.py, profiles/r4_words_stamps.txt)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t row_bytes = a.ld * 8u, row_pad = (row_bytes + 16u + 1023u) & ~1023u;  // row, then the +inf cell
  // NEG: word-end score of every word (unpruned, +inf: none), [2] by frame parity, behind the row buffers
  const uint32_t n_words = a.net.n_words, we_pad = (n_words * 8u + 1023u) & ~1023u;
  double* we_all = reinterpret_cast<double*>(rows_lds + 2u * row_pad);

  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x;
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;

  // ---- the lane's words: slot tid + k * nt of the host's order ---------------------------------------------------------------
  uint32_t word[NW], info[NW], o_end[NW], st[NW][NPA], kind[NW];
  uint64_t real[NW];  // lanes of the group that hold a word
  double sc[NW][NPA];
  uint32_t bk[NW][NPA];
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const uint32_t w = a.words.order[tid + (uint32_t)k * nt];
    const bool in = w != 0xFFFFFFFFu;
    word[k] = w;
    info[k] = in ? a.words.info[w] : 0u;
    o_end[k] = in ? a.net.word_end_slot[w] : 0u;  // original index of the word's last position
    const uint2 s4 = in ? a.words.states[w] : make_uint2(0u, 0u);
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const uint32_t half = p < 2 ? s4.x : s4.y;
      // byte offset of the emission cost in the row; padding lanes: the +inf cell behind the row
      st[k][p] = in ? ((half >> (16 * (p & 1))) & 0xFFFFu) * 8u : row_bytes;
      sc[k][p] = kInfF;
      bk[k][p] = 0u;
    }
    if (w == 0) sc[k][0] = 0.0;  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
    real[k] = __ballot(in);
    const uint32_t flags = info[k] & (7u | kWSilWord | kWFirstSil | kWSilStates);
    kind[k] = real[k] == 0 ? kGSkip : __all(!in || flags == (uint32_t)L) ? kGPlain : __all(!in || (info[k] & 7u) == 1u) ? kGSingle : kGGeneral;
  }
  // The host puts a general group into slot row k = 0 of a wave of its own (srgpu_api.cpp), so only k = 0 carries the registers of
  // a fourth position and the general code.  (A general group anywhere else -- not something sr_lexicon_create produces -- sends the
  // utterance to the replay kernel.)
  bool misplaced = false;
#pragma unroll
  for (int k = 1; k < NW; k++) misplaced |= kind[k] == kGGeneral;
  if (!GEN) misplaced |= kind[0] == kGGeneral;
  if (tid < 12) e_first[tid] = 0xFFFFFFFFu;
  if (tid < 3) { c_best[tid] = kInfF; c_we[tid] = kInfF; c_widx[tid] = 0xFFFFFFFFu; }
  if (tid == 12) *s_bad = 0;
  // (Compiling the frame loop a second time for waves whose groups are all plain -- straight-line code, no dispatch on the group
  // kind -- was measured too: 2.30 ms against 2.22, 126 registers and scratch against 119.)
  // (Round 3: fetching rows two frames ahead into THREE buffers was measured, 3.31 ms per step against 2.99 with the same geometry,
  // and the third buffer costs the second workgroup per CU, which is worth 2.99 -> 2.24 ms.  Round 4 gets the longer flight time
  // from two buffers by issuing the copy right behind the barrier, see the frame loop.)
  if (tid < 2) *reinterpret_cast<double*>(rows_lds + tid * row_pad + row_bytes) = kInfF;
  const bool init_is_end = a.words.init_is_end;
  if constexpr (NEG) {  // "frame 0": the initial hypothesis (word 0, score 0) is the one word end, if it is one
    for (uint32_t i = tid; i < 2u * (we_pad >> 3); i += nt) we_all[i] = kInfF;
  }
  double m_we = init_is_end ? 0.0 : kInfF;  // minimum over the word ends that survived the previous frame (uniform)
  if (tid == 0) { a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0; }
  __syncthreads();
  if (tid < 4 && init_is_end) e_first[tid] = 0;  // "frame 0": the initial hypothesis is a word end of index 0 in every class
  if (NEG && tid == 0 && init_is_end) we_all[0] = 0.0;  // (buffer 0 = frame 0; published by the barrier below)
  double limit_prev = kInfF;  // NEG: the previous frame's prune limit (uniform)

  const uint32_t n_full = row_bytes >> 10, tail_bytes = row_bytes & 1023u;  // whole 1 KB pieces of a row; what the last, short piece holds
  auto issue_row = [&](uint32_t frame /* 1-based */) {  // row of `frame` -> buffer frame & 1; every wave copies its share of the pieces
    const unsigned char* src = reinterpret_cast<const unsigned char*>(row0 + (uint64_t)(frame - 1) * a.ld) + lane * 16u;
    unsigned char* dst = rows_lds + (frame & 1u) * row_pad;
    uint32_t piece = wave;  // (scalar loop: one s_add per piece for the source, one for M0)
    for (; piece < n_full; piece += n_waves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024u), 16, 0, 0);
    if (piece == n_full && lane * 16u < tail_bytes)  // (a row is a multiple of 64 bytes: the last piece may be short)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024u),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024u), 16, 0, 0);
  };
  if (T > 0) { issue_row(1); __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)
  __syncthreads();
  if (T > 1) issue_row(2);  // (waited for before the barrier of frame 1)

  uint64_t bad = misplaced ? ~0ull : 0ull;  // lanes that met an emission cost that is not >= 0 (scalar mask, looked at after the last frame)
  // the lane that may own traceback[t]: it held the word-end minimum of frame t; settled by the atomic, written one barrier later
  bool pend = false;
  uint32_t pend_o = 0, pend_w = 0, pend_b = 0;
  double pend_v = 0.0;
  bool prev_alive = true;  // frame 0: traceback[0] is written above
  auto flush_pending = [&](const uint32_t t_prev) {  // t_prev >= 1, after the barrier that follows frame t_prev's atomics
    if (__any(pend)) {  // wave-uniform, about one wave per frame
      if (pend && c_widx[t_prev % 3u] == pend_o) {
        a.tb_score[tb0 + t_prev] = pend_v; a.tb_word[tb0 + t_prev] = (uint16_t)pend_w; a.tb_bkp[tb0 + t_prev] = (uint16_t)pend_b;
      }
      pend = false;
    }
    // no word end survived: Book(inf, 0, 0), Recognizer.cpp:118,191
    if (!prev_alive && tid == 0) { a.tb_score[tb0 + t_prev] = kInfF; a.tb_word[tb0 + t_prev] = 0; a.tb_bkp[tb0 + t_prev] = 0; }
  };

  uint32_t r = 1, r_prev = 0;  // t % 3, (t - 1) % 3
  for (uint32_t t = 1; t <= T; t++) {
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;
    const unsigned char* row_l = rows_lds + (t & 1u) * row_pad;
    const uint64_t we_in = m_we != kInfF ? ~0ull : 0ull;  // uniform: a word end survived the previous frame -- else every boundary candidate is +inf
    // the collapsed word-boundary candidate of a plain word before its emission cost: cur_hyp->score + word_penalty + tdp
    // (Recognizer.cpp:135-140); forward into position 0, skip into position 1 -- the same in every lane
    const double cb_f = (m_we + wp_word) + tf, cb_s = (m_we + wp_word) + ts;

    // ---- A: the new hypotheses of the lane's words, in place -- position p's sources are p, p - 1, p - 2: descending p -------
    // tie[k][p]: lanes where the boundary candidate of word k's position p (0 or 1) ties with the in-word minimum (scalar masks)
    uint64_t tie[NW][2];
    double my_best = kInfF, my_we = kInfF;
    double e[NW][NPA];
#pragma unroll
    for (int k = 0; k < NW; k++) {
      tie[k][0] = tie[k][1] = 0;
      if (kind[k] == kGSkip) continue;
#pragma unroll
      for (int p = 0; p < NPA; p++)
        if ((GEN && k == 0 && kind[k] == kGGeneral) || (kind[k] == kGPlain && p < L) || p == 0) e[k][p] = *reinterpret_cast<const double*>(row_l + st[k][p]);
    }
    if constexpr (NEG) {
      // ---- NEG: exact under negative emission costs (file header).  One word of the lane; PLAIN: L positions, no silence flags (every
      // penalty a scalar); NP = positions compiled in (L, 1 for a group of one-position words, 4 for a general group) -------------
      const double* we_prev = we_all + ((t - 1u) & 1u) * (we_pad >> 3);
      double* we_cur = we_all + (t & 1u) * (we_pad >> 3);
      auto neg_word = [&](auto plain_tag, auto np_tag, const int k) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
        constexpr int NP = decltype(np_tag)::value;
        uint32_t inf_k = info[k];
        asm volatile("" : "+v"(inf_k));  // (nothing derived from the flags is to be hoisted out of the frame loop: registers)
        const uint32_t n = PLAIN ? (uint32_t)L : (inf_k & 7u);
        const bool in = PLAIN ? (bool)((real[k] >> lane) & 1ull) : n != 0u;
        const bool sil_word = PLAIN ? false : (bool)(inf_k & kWSilWord), first_sil = PLAIN ? false : (bool)(inf_k & kWFirstSil);
        const double wp_l = sil_word ? 0.0 : wp_word;
        const double base_we = m_we + wp_l;
        const double e0 = e[k][0], e1 = NP >= 2 ? e[k][NP >= 2 ? 1 : 0] : 0.0;
        const double old0 = sc[k][0], old1 = NP >= 2 ? sc[k][NP >= 2 ? 1 : 0] : kInfF;
        const uint32_t ob0 = bk[k][0], ob1 = NP >= 2 ? bk[k][NP >= 2 ? 1 : 0] : 0u;
        // positions >= 2: skip, forward, loop in source order through the early-out
#pragma unroll
        for (int p = NP - 1; p >= 2; p--) {
          const bool valid = (uint32_t)p < n, end = (uint32_t)p + 1u == n;
          const bool sil = PLAIN ? false : (bool)((inf_k >> (8 + p)) & 1u);
          const double t_loop = sil ? tf : tl, t_skip = sil ? tf : ts;  // TdpModel.cpp:19-29, keyed on the destination state
          const double ep = e[k][p];
          WMerge mg;
          mg.offer(sc[k][p - 2] + t_skip, ep, bk[k][p - 2]);
          mg.offer(sc[k][p - 1] + tf, ep, bk[k][p - 1]);
          mg.offer(end ? kInfF : sc[k][p] + t_loop, ep, bk[k][p]);
          sc[k][p] = valid ? mg.score : kInfF; bk[k][p] = mg.bkp;
        }
        // positions 1 and 0: the collapsed boundary candidate, exact while the word's entry emission costs are >= 0
        const bool sil0 = PLAIN ? false : (bool)((inf_k >> 8) & 1u), sil1 = PLAIN ? false : (bool)((inf_k >> 9) & 1u);
        const bool end0 = n == 1u, end1 = n == 2u, valid1 = n >= 2u;
        const double pre_loop0 = end0 ? kInfF : old0 + (sil0 ? tf : tl);
        const double pre_fwd1 = valid1 ? old0 + tf : kInfF, pre_loop1 = (valid1 && !end1) ? old1 + (sil1 ? tf : tl) : kInfF;
        const double tb1 = first_sil ? tf : ts;  // tdp(first_state, init + 1): position 1, or the dead slot of a one-position word
        double v1 = kInfF, dead = kInfF;
        uint32_t b1 = 0;
        if (NP >= 2) {
          v1 = pre_fwd1 + e1; b1 = ob0;
          const double s0 = pre_loop1 + e1;
          b1 = s0 < v1 ? ob1 : b1;
          v1 = dmin(v1, s0);
          const double n_b = (base_we + tb1) + e0;  // scored with position 0's emission (Recognizer.cpp:136,148-151)
          tie[k][1] = __ballot(valid1 && n_b == v1) & real[k] & we_in;
          b1 = n_b < v1 ? bkp_new : b1;
          v1 = valid1 ? dmin(v1, n_b) : kInfF;
        }
        double v0 = pre_loop0 + e0;
        uint32_t b0 = ob0;
        {
          const double n_b = (base_we + tf) + e0;
          tie[k][0] = __ballot(in && n_b == v0) & real[k] & we_in;
          b0 = n_b < v0 ? bkp_new : b0;
          v0 = dmin(v0, n_b);
          if (!PLAIN) dead = end0 ? (base_we + tb1) + e0 : kInfF;  // feeds best_score only (Recognizer.cpp:139,155)
          if (!PLAIN && end0) b0 = v0 < kInfF ? bkp_new : 0u;
        }
        // ... else the lane replays the boundary loop for its word
        const bool slow = in && (!(e0 >= 0.0) || (valid1 && !(e1 >= 0.0)));
        const uint64_t slow_mask = __ballot(slow);
        if (slow_mask) {  // wave-uniform
          WMerge m0, m1;  // position 0; position 1 (n >= 2) or the dead slot (n == 1)
          // ONE pass over the live word ends in hypothesis order; a lane's in-word sources take their turn when the pass reaches the
          // first word end that is not before the lane's word (a wave-uniform branch, taken once per slow lane at most)
          bool inword_due = slow;
          constexpr int kAhead = 8;  // array reads in flight: a dependent LDS round trip per 64 words cost a third of a frame at W = 1334
#pragma unroll 1
          for (uint32_t base0 = 0; base0 < n_words; base0 += 64u * kAhead) {
            double svs[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; j++) {
              const uint32_t vi = base0 + 64u * j + lane;
              svs[j] = vi < n_words ? we_prev[vi] : kInfF;
            }
#pragma unroll
            for (int j = 0; j < kAhead; j++) {
              const double sv = svs[j];
              const uint32_t base = base0 + 64u * j;
              uint64_t live = __ballot(sv != kInfF && !(sv > limit_prev));  // the previous frame's survivors (:194-196)
              while (live) {
                const uint32_t i = (uint32_t)__builtin_ctzll(live);
                live &= live - 1ull;
                const double s_e = readlane_d(sv, i);
                const uint32_t v = base + i;
                const bool now = inword_due && v >= word[k];
                if (__ballot(now)) {  // the in-word sources, ascending, before this word end
                  m0.offer(now ? pre_loop0 : kInfF, e0, ob0);
                  m1.offer(now ? pre_fwd1 : kInfF, e1, ob0);
                  m1.offer(now ? pre_loop1 : kInfF, e1, ob1);
                  inword_due = inword_due && !now;
                }
                const double c = s_e + wp_l;
                m0.offer(slow ? c + tf : kInfF, e0, bkp_new);
                m1.offer(slow ? c + tb1 : kInfF, e0, bkp_new);
              }
            }
          }
          m0.offer(inword_due ? pre_loop0 : kInfF, e0, ob0);  // (no live word end at or behind the lane's word)
          m1.offer(inword_due ? pre_fwd1 : kInfF, e1, ob0);
          m1.offer(inword_due ? pre_loop1 : kInfF, e1, ob1);
          if (slow) {
            v0 = m0.score; b0 = m0.score < kInfF ? m0.bkp : 0u;
            if (valid1) { v1 = m1.score; b1 = m1.score < kInfF ? m1.bkp : 0u; } else dead = m1.score;
          }
          tie[k][0] &= ~slow_mask; tie[k][1] &= ~slow_mask;
        }
        v0 = in ? v0 : kInfF;
        sc[k][0] = v0; bk[k][0] = b0;
        if (NP >= 2) { sc[k][NP >= 2 ? 1 : 0] = v1; bk[k][NP >= 2 ? 1 : 0] = b1; }
        double w_end = kInfF;  // the word's last position
#pragma unroll
        for (int p = 0; p < NP; p++) {
          my_best = dmin(my_best, sc[k][p]);
          if (PLAIN ? p == L - 1 : (uint32_t)p + 1u == n) w_end = sc[k][p];
        }
        my_best = dmin(my_best, dead);
        my_we = dmin(my_we, w_end);
        if (in) we_cur[word[k]] = w_end;  // unpruned; next frame's replay applies this frame's limit
      };
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (kind[k] == kGPlain) neg_word(std::true_type{}, std::integral_constant<int, L>{}, k);
        else if (kind[k] == kGSingle) neg_word(std::false_type{}, std::integral_constant<int, 1>{}, k);
        else if (GEN && k == 0 && kind[k] == kGGeneral) neg_word(std::false_type{}, std::integral_constant<int, NPA>{}, k);
      }
    } else {
#pragma unroll
    for (int k = 0; k < NW; k++) {
      if (kind[k] == kGPlain) {
        // ---- plain: L positions, the last one the word end; penalties are the scalars tl, tf, ts, wp_word -------------------
#pragma unroll
        for (int p = L - 1; p >= 0; p--) {
          const double ep = e[k][p];
          bad |= __ballot(!(ep >= 0.0));
          double v = kInfF;
          uint32_t b = 0;
          if (p >= 2) {  // skip, forward: in source order, a later one must be strictly better
            const double s2 = (sc[k][p - 2] + ts) + ep, s1 = (sc[k][p - 1] + tf) + ep;
            b = s1 < s2 ? bk[k][p - 1] : bk[k][p - 2];
            v = dmin(s2, s1);
          } else if (p == 1) {
            v = (sc[k][0] + tf) + ep;
            b = bk[k][0];
          }
          if (p != L - 1) {  // loop, unless the source is the word end (word-end hypotheses are expanded across the boundary only, :130-158)
            const double s0 = (sc[k][p] + tl) + ep;
            b = s0 < v ? bk[k][p] : b;
            v = dmin(v, s0);
          }
          if (p <= 1) {  // the boundary candidate, scored with position 0's emission (Recognizer.cpp:136,148-151)
            const double n_b = (p == 1 ? cb_s : cb_f) + e[k][0];
            tie[k][p] = __ballot(n_b == v) & real[k] & we_in;
            b = n_b < v ? bkp_new : b;
            v = dmin(v, n_b);
          }
          sc[k][p] = v; bk[k][p] = b;
          my_best = dmin(my_best, v);
          if (p == L - 1) my_we = dmin(my_we, v);
        }
      } else if (kind[k] == kGSingle) {
        // ---- one-position words: the boundary candidate (position 0 is the word end: no loop) and the dead position-1
        // hypothesis, which still feeds best_score (Recognizer.cpp:139,155) ------------------------------------------------------
        uint32_t inf_k = info[k];
        asm volatile("" : "+v"(inf_k));  // (nothing derived from the flags is to be hoisted out of the frame loop: registers)
        const double ep = e[k][0];
        bad |= __ballot(!(ep >= 0.0));
        const double base_we = m_we + ((inf_k & kWSilWord) ? 0.0 : wp_word);
        const double n_b = (base_we + tf) + ep;
        const double dead = (base_we + ((inf_k & kWFirstSil) ? tf : ts)) + ep;
        sc[k][0] = n_b; bk[k][0] = n_b < kInfF ? bkp_new : 0u;
        my_best = dmin(my_best, dmin(n_b, dead));
        my_we = dmin(my_we, n_b);
      } else if (GEN && k == 0 && kind[k] == kGGeneral) {
        // ---- general: 0 .. 4 positions per lane, silence word, silence states -------------------------------------------------
        uint32_t inf_k = info[k];
        asm volatile("" : "+v"(inf_k));
        const uint32_t n = inf_k & 7u;
        const bool sil_word = inf_k & kWSilWord, first_sil = inf_k & kWFirstSil;
        const double base_we = m_we + (sil_word ? 0.0 : wp_word);
#pragma unroll
        for (int p = NPA - 1; p >= 0; p--) {
          const bool valid = (uint32_t)p < n, end = (uint32_t)p + 1u == n;
          const bool sil = (inf_k >> (8 + p)) & 1u;
          const double t_loop = sil ? tf : tl, t_skip = sil ? tf : ts;  // TdpModel.cpp:19-29, keyed on the destination state
          const double ep = e[k][p];
          bad |= __ballot(valid && !(ep >= 0.0));
          double v = kInfF;
          uint32_t b = 0;
          if (p >= 2) {
            const double s2 = (sc[k][p - 2] + t_skip) + ep, s1 = (sc[k][p - 1] + tf) + ep;
            b = s1 < s2 ? bk[k][p - 1] : bk[k][p - 2];
            v = dmin(s2, s1);
          } else if (p == 1) {
            v = (sc[k][0] + tf) + ep;
            b = bk[k][0];
          }
          {
            const double s0 = end ? kInfF : (sc[k][p] + t_loop) + ep;
            b = s0 < v ? bk[k][p] : b;
            v = dmin(v, s0);
          }
          if (p <= 1) {
            const bool b_skip = p == 1 && !first_sil;
            const double n_b = (base_we + (b_skip ? ts : tf)) + e[k][0];
            tie[k][p] = __ballot(valid && n_b == v) & we_in;
            b = n_b < v ? bkp_new : b;
            v = dmin(v, n_b);
            if (p == 0) {
              const double dead = (base_we + (first_sil ? tf : ts)) + ep;
              my_best = dmin(my_best, n == 1u ? dead : kInfF);
            }
          }
          v = valid ? v : kInfF;
          sc[k][p] = v; bk[k][p] = b;
          my_best = dmin(my_best, v);
          my_we = dmin(my_we, end ? v : kInfF);
        }
      }
    }
    }

    // ---- B: block minima through LDS ds_min_f64 cells, fed by one lane per row of 16 ---------------------------------------
    my_best = row_min_dpp(my_best);
    my_we = row_min_dpp(my_we);
    if ((lane & 15u) == 0) publish_min2_f64_lds(&c_best[r], my_best, &c_we[r], my_we);  // (atomics + their wait: dpp_util.h)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next row have landed; the barrier publishes them
    __syncthreads();
    // Row t + 2 goes into the buffer frame t has just finished with: every wave read its emission costs (phase A) before this
    // barrier and phase C reads none, so the copy has phase C and phases A and B of frame t + 1 to land in -- a whole frame, where
    // round 3 issued row t + 1 at the top of frame t and gave it phases A and B only (stamps: 620 cycles of a 7 650-cycle frame
    // in the vmcnt wait and most of the 1 700 at the barrier at configs[2], profiles/r4_words_stamps.txt).  Still two row buffers.
    if (t + 2 <= T) issue_row(t + 2);

    // ---- C: prune, word-end bookkeeping ------------------------------------------------------------------------------------
    const double best = c_best[r], we = c_we[r];
    const double limit = best + thr;
    const bool we_alive = !(we > limit) && we != kInfF;
    if constexpr (NEG) limit_prev = limit;
    {
      uint64_t tie_any = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) tie_any |= tie[k][0] | tie[k][1];
      if (tie_any) {  // rare: the boundary source came first where the first minimal word end of the class precedes the word
        const uint32_t* ef_prev = e_first + 4 * r_prev;  // written in phase C of frame t - 1: complete since this frame's barrier
#pragma unroll
        for (int k = 0; k < NW; k++) {
          const uint32_t n = info[k] & 7u, base = o_end[k] - (n - 1u);
          const bool sil_word = info[k] & kWSilWord, first_sil = info[k] & kWFirstSil;
#pragma unroll
          for (int p = 0; p < 2 && p < NPA; p++) {
            const bool b_skip = p == 1 && !first_sil;
            const uint32_t cls = (sil_word ? 0u : 2u) + (b_skip ? 1u : 0u);
            if ((tie[k][p] >> lane & 1ull) && ef_prev[cls] < base) bk[k][p] = bkp_new;  // the in-word candidate had to be strictly better
          }
        }
      }
    }
    if (t > 1) flush_pending(t - 1);
    m_we = we_alive ? we : kInfF;
    prev_alive = we_alive;
    const double near = m_we + (fabs(m_we) + fabs(wp_word) + fabs(tf) + fabs(ts) + 1.0) * 1e-9;
    double v_end[NW];
    uint32_t b_end[NW];
    uint64_t any_near = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      v_end[k] = kInfF; b_end[k] = 0;
      if (kind[k] == kGSkip) continue;
#pragma unroll
      for (int p = 0; p < NPA; p++) {
        if ((k > 0 && p >= L) || (kind[k] == kGPlain && p >= L) || (kind[k] == kGSingle && p >= 1)) continue;
        double v = sc[k][p];
        if (v > limit) v = kInfF;  // :194-196
        sc[k][p] = v;
      }
      if (kind[k] == kGPlain) {
        v_end[k] = sc[k][L - 1]; b_end[k] = bk[k][L - 1];
      } else if (kind[k] == kGSingle) {
        v_end[k] = sc[k][0]; b_end[k] = bk[k][0];
      } else if (k == 0) {
        const uint32_t n = info[k] & 7u;
#pragma unroll
        for (int p = 0; p < NPA; p++)
          if ((uint32_t)p + 1u == n) { v_end[k] = sc[k][p]; b_end[k] = bk[k][p]; }
      }
      any_near |= __ballot(v_end[k] <= near);
    }
    if (we_alive && any_near) {  // the wave that holds the minimum (about one lane of the block)
      uint32_t* ef_nxt = e_first + 4 * r;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const double v = v_end[k];
        if (v <= near) {
          const uint32_t o = o_end[k];
          const bool is_min = v == m_we;
          if (is_min) {  // traceback[t] = the FIRST minimal surviving word end (:199-205): settled by the atomic, written after the next barrier
            // (a lane may hold several of them: it keeps the one with the smallest original index, the only one that can win)
            if (!pend || o < pend_o) { pend_o = o; pend_w = word[k]; pend_b = b_end[k]; pend_v = v; }
            pend = true;
          }
          // ... and the first word end per boundary class whose candidate (score + word penalty + tdp) equals the minimum's after
          // rounding; all five atomic minima in one statement (dpp_util.h), 0xFFFFFFFF = nothing to contribute
          const uint32_t none = 0xFFFFFFFFu;
          lds_min5_u32(&c_widx[r], is_min ? o : none, ef_nxt,
                       v + 0.0 + tf == m_we + 0.0 + tf ? o : none, v + 0.0 + ts == m_we + 0.0 + ts ? o : none,
                       v + wp_word + tf == m_we + wp_word + tf ? o : none, v + wp_word + ts == m_we + wp_word + ts ? o : none);
        }
      }
    }
    // resets, each one barrier before the cell's next atomics and after its last reads: the minima of frame t + 2 (last read
    // in phase C of t - 1), the index cells of frame t + 1 (last read in phase C of t - 1 too: flush and tie patch of t - 2's)
    {
      const uint32_t r_next = r == 2 ? 0u : r + 1u;
      if (tid == 0) { c_best[r_prev] = kInfF; c_we[r_prev] = kInfF; c_widx[r_next] = 0xFFFFFFFFu; }  // (t + 2) % 3 == (t - 1) % 3
      if (tid >= 4 && tid < 8) e_first[4 * r_next + (tid - 4)] = 0xFFFFFFFFu;
      r_prev = r; r = r_next;
    }
  }
  __syncthreads();
  if (T > 0) flush_pending(T);

  // the premise failed somewhere (a negative or NaN emission cost): hand the utterance to the replay variant
  if (bad) *s_bad = 1;
  __threadfence();
  __syncthreads();
  if (*s_bad) {  // workgroup-uniform
    if (tid == 0) { atomicOr(&a.out_flags[u], kFlagReplay); a.out_count[u] = 0; }
    return;
  }

  // ---- traceback (Recognizer.cpp:222-231; guarded walk: traceback.h) ---------------------------------------------------
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

static size_t words_smem(uint32_t ld, uint32_t n_words = 0, bool neg = false) {
  return kWordsCellBytes + 2 * (((size_t)ld * 8 + 16 + 1023u) & ~(size_t)1023u) + (neg ? 2 * (((size_t)n_words * 8 + 1023u) & ~(size_t)1023u) : 0);
}
static constexpr size_t kLdsPerWorkgroup = 160 * 1024;

bool decode_words_applies(const DecodeArgs& a) { return a.words.info != nullptr && words_smem(a.ld) <= kLdsPerWorkgroup; }
uint32_t decode_words_max_words() { return 3 * 1024; }

hipError_t launch_decode_words(const DecodeArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  if (!decode_words_applies(a)) return hipErrorInvalidValue;
  const uint32_t nw = a.words.nw, nt = a.words.nt;
  if (nw < 1 || nw > 3 || nt == 0 || nt > 1024 || nt % 64) return hipErrorInvalidValue;
  // the model's emission costs can be negative: the variant that replays the reference's early-out (NEG), where its word-end
  // arrays fit the LDS beside the two score rows; else the fast variant, which flags what it cannot do for the replay kernel
  const bool neg = a.exact_negative && words_smem(a.ld, a.net.n_words, true) <= kLdsPerWorkgroup;
  const size_t smem = words_smem(a.ld, a.net.n_words, neg);
  auto go = [&](auto kernel) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(a.n_utts), dim3(nt), smem, stream, a);
    return hipGetLastError();
  };
#define SR_WORDS_N(Lv, Gv, Nv, Tv) do { if (nw == 1) return go(decode_words_kernel<1, Lv, Gv, Nv, Tv>); if (nw == 2) return go(decode_words_kernel<2, Lv, Gv, Nv, Tv>); return go(decode_words_kernel<3, Lv, Gv, Nv, Tv>); } while (0)
#define SR_WORDS_G(Lv, Gv) do { if (neg && nt <= 512) SR_WORDS_N(Lv, Gv, true, 512); if (neg) SR_WORDS_N(Lv, Gv, true, 1024); SR_WORDS_N(Lv, Gv, false, 1024); } while (0)
#define SR_WORDS(Lv) do { if (a.words.has_general) SR_WORDS_G(Lv, true); SR_WORDS_G(Lv, false); } while (0)
  if (a.words.plain_len == 2) SR_WORDS(2);
  if (a.words.plain_len == 4) SR_WORDS(4);
  SR_WORDS(3);
#undef SR_WORDS
#undef SR_WORDS_G
#undef SR_WORDS_N
}

}  // namespace srgpu
