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
// This file holds the GENERAL kernel: slots in reference order, every per-slot decision taken per lane, the
// sequential boundary replay inline.  Production launches run viterbi_fast.hip's type-sorted kernel first (same
// results, ~5x fewer instructions per frame) and this one, as decode_kernel<.., REPLAY = true>, only for the
// utterances the fast kernel hands back (out_flags bit 1: a negative emission cost was seen).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "traceback.h"

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
    const double n = pre_am + am;
    const bool take = !(pre_am > score) && (score > n);  // `continue` on the early-out, then strict improvement
    score = take ? n : score;
    bkp = take ? cand_bkp : bkp;
  }
};

// (value, index) lexicographic minimum across the wave: every lane ends with the result
__device__ inline void wave_min_idx(double& v, uint32_t& idx) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    const double ov = __hiloint2double(hi, lo);
    const uint32_t oi = __shfl_xor(idx, m);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
__device__ inline double wave_min(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    const double ov = __hiloint2double(hi, lo);
    v = ov < v ? ov : v;
  }
  return v;
}

// Sequential replay of the word-boundary loop (Recognizer.cpp:133-158) for ONE target slot whose
// candidates involve a negative emission cost: sources in ascending slot order -- word ends of the
// words before this one, the in-word sources, then the remaining word ends.  Rare; reached only through a
// wave-uniform branch.
__device__ inline Merge replay_boundary(const uint32_t* word_end_slot, uint32_t W, const double* sc,
                                                           const uint16_t* bk, uint32_t p, bool pos1, bool has_loop,
                                                           bool in_word, double wp, double t_b, double am_b, double am,
                                                           double t_fwd, double t_loop, uint32_t bkp_new) {
  Merge mg;
  const uint32_t base = pos1 ? p - 1 : p;  // slot of position 0 of this word
  uint32_t v = 0;
  for (; v < W; v++) {
    const uint32_t e = word_end_slot[v];
    if (e >= base) break;
    const double s = sc[e];
    if (s != kInf) mg.offer((s + wp) + t_b, am_b, bkp_new);
  }
  if (in_word) {
    if (pos1) mg.offer(sc[p - 1] + t_fwd, am, bk[p - 1]);
    if (has_loop) mg.offer(sc[p] + t_loop, am, bk[p]);
  }
  for (; v < W; v++) {
    const double s = sc[word_end_slot[v]];
    if (s != kInf) mg.offer((s + wp) + t_b, am_b, bkp_new);
  }
  return mg;
}

// REPLAY = false: the fast variant.  It carries no sequential-replay code; if an entry slot ever meets a
// negative emission cost it raises out_flags bit 1 for the utterance and the whole workgroup stops.
// REPLAY = true: launched right after on the same stream; workgroups of unflagged utterances exit at
// once, flagged utterances are decoded again from frame 1 with the replay inline (exact, slower).
template <int NT, int SPT, bool REPLAY>
__global__ __launch_bounds__(NT) void decode_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kWavesPerWg = NT / 64;
  const uint32_t P = a.net.n_slots;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr uint32_t PP = NT * SPT;                                // slot arrays are padded to the thread grid
  double* sc = reinterpret_cast<double*>(smem);                    // [PP] hypothesis scores
  double* am_l = sc + PP;                                          // [PP] this frame's emission cost per slot (for position-1 slots' neighbours)
  double* red_best = am_l + PP;                                    // [16]
  double* red_we = red_best + 16;                                  // [16]
  uint32_t* red_idx = reinterpret_cast<uint32_t*>(red_we + 16);    // [16]
  uint32_t* e_first = red_idx + 16;                                // [2][4] first word-end slot per class, by frame parity
  uint32_t* bail = e_first + 8;                                    // [1] fast variant: an entry slot needs the replay
  uint16_t* bk = reinterpret_cast<uint16_t*>(e_first + 12);         // [PP] back pointers (start frame of the word)

  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x;
  if (REPLAY && !a.force_general && !(a.out_flags[u] & kFlagReplay)) return;  // wave-uniform: nothing to redo for this utterance
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;  // traceback[0] of this utterance
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;

  // ---- static per-slot constants: info | first_state of the word (for position-1 slots) ------------
  uint32_t info[SPT], pk[SPT];  // pk = bkp of the hypothesis being built
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = tid + i * NT;
    info[i] = p < P ? a.net.slot_info[p] : 0u;
    pk[i] = 0;
    sc[p] = kInf; bk[p] = 0;
  }
  if (tid < 8) e_first[tid] = 0xFFFFFFFFu;
  if (tid == 8) *bail = 0;
  __syncthreads();
  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
  const bool init_is_end = a.net.slot_info[0] & kSlotEnd;
  double m_we = init_is_end ? 0.0 : kInf;  // min over live word-end slots of the previous frame
  if (tid == 0) {
    sc[0] = 0.0;
    a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0;  // traceback[0] = Book(0.0,0,0,0), :118
  }
  if (tid < 4 && init_is_end) e_first[4 + tid] = 0;  // frame t = 1 reads parity buffer 1
  // Emission costs are gathered TWO frames ahead into two register sets (set f&1 holds frame f, frames are
  // 1-based) and mirrored into am_l one frame ahead, so neither the frame body nor its barriers ever wait on
  // HBM: the gathers of frame t+2 are issued at the top of frame t and first touched in phase C of frame t+1.
  double am_s0[SPT], am_s1[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const uint32_t p = tid + i * NT;
    am_s1[i] = T > 0 ? row0[info[i] & 0xFFFFu] : 0.0;  // padding slots carry state 0: a valid address, result unused
    am_l[p] = am_s1[i];
    am_s0[i] = T > 1 ? row0[a.ld + (info[i] & 0xFFFFu)] : 0.0;
  }
  __syncthreads();
  uint32_t slow_taken = 0;

  // one frame; returns true when the fast variant has to hand the utterance to the replay variant
  auto frame = [&](const uint32_t t, double (&am_issue)[SPT], double (&am_consume)[SPT]) -> bool {
    const uint32_t* ef_cur = e_first + 4 * (t & 1);
    uint32_t* ef_nxt = e_first + 4 * ((t + 1) & 1);
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;  // merge_hypothesis(.., t - 1, ..) truncated to uint16 (:154, Recognizer.hpp:79)
    if (t + 2 <= T) {
      const double* rown = row0 + (uint64_t)(t + 1) * a.ld;  // frame t+2
#pragma unroll
      for (int i = 0; i < SPT; i++) am_issue[i] = rown[info[i] & 0xFFFFu];
    }

    // ---- A: build the new hypotheses in registers ------------------------------------------------
    // Branch-free: every slot makes the same five offers in ascending source order -- [boundary], skip,
    // forward, loop, [boundary] -- where a candidate that does not exist for this slot is +inf, which
    // Merge::offer ignores exactly like the reference ignores an absent hypothesis (:127-129).
    double nv[SPT];
    double my_best = kInf, my_we = kInf;
    uint32_t my_we_idx = 0xFFFFFFFFu;
    bool any_slow = false;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      uint32_t p = tid + i * NT;
      const bool live = p < P;
      asm volatile("" : "+v"(p));  // keep p*8 / p*2 address math out of loop-invariant registers
      // launder the slot constants: otherwise LICM hoists every derived per-slot value (tdp selects, word
      // penalty, class: ~12 VGPRs per slot) out of the frame loop and occupancy drops to 1 wave/SIMD
      uint32_t inf = info[i];
      asm volatile("" : "+v"(inf));
      const double am = am_l[p];
      const bool pos0 = inf & kSlotPos0, pos1 = inf & kSlotPos1, entry = pos0 || pos1;
      const bool sil_state = inf & kSlotSilState;
      const double t_loop = sil_state ? tf : tl, t_fwd = tf, t_skip = sil_state ? tf : ts;
      // unconditional LDS reads at clamped addresses; the selects below discard what does not apply
      const uint32_t p1 = p - (pos0 ? 0u : 1u), p2 = p - (entry ? 0u : 2u);
      const double s0 = sc[p], s1 = sc[p1], s2 = sc[p2], a1 = am_l[p1];
      const uint32_t b_loop_ = bk[p], b_fwd_ = bk[p1], b_skip_ = bk[p2];
      const double c_skip = entry ? kInf : s2 + t_skip;
      const double c_fwd = pos0 ? kInf : s1 + t_fwd;
      const double c_loop = (inf & kSlotEnd) ? kInf : s0 + t_loop;  // word ends do not expand in-word (:131)
      // word-boundary candidate (positions 0 and 1 only)
      const double am_b = pos1 ? a1 : am;  // emission of the word's position 0, even when entering position 1 (:136,148-151)
      const double wp = (inf & kSlotSilWord) ? 0.0 : wp_word;
      const bool b_skip = pos1 && !(inf & kSlotFirstSil);  // tdp(first_state, init + 1)
      const double t_b = b_skip ? ts : tf;
      const uint32_t cls = ((inf & kSlotSilWord) ? 0u : 2u) + (b_skip ? 1u : 0u);
      const double c_b = entry ? (m_we + wp) + t_b : kInf;  // cur_hyp->score + word_penalty + tdp, :140
      const uint32_t e_b = ef_cur[cls];                     // first word-end slot attaining c_b
      const bool b_first = e_b < p1;                        // is that source visited before the in-word ones? (p1 == p for position 0)
      Merge mg;
      mg.offer(b_first ? c_b : kInf, am_b, bkp_new);
      mg.offer(c_skip, am, b_skip_);
      mg.offer(c_fwd, am, b_fwd_);
      mg.offer(c_loop, am, b_loop_);
      mg.offer(b_first ? kInf : c_b, am_b, bkp_new);
      double extra = kInf;  // one-position word: boundary candidates with init = 1 land in a slot past the word's
                            // end (Recognizer.cpp:139); it never expands but feeds best_score (:155)
      if (live && (inf & kSlotSingle)) extra = ((m_we + wp) + ((inf & kSlotFirstSil) ? tf : ts)) + am;
      // the collapsed boundary candidate is only exact while the pre-AM early-out (:143) is inert
      const bool slow = live && (entry && (am_b < 0.0 || am < 0.0));
      any_slow |= slow;
      nv[i] = live ? mg.score : kInf; pk[i] = mg.bkp | (slow ? 0x80000000u : 0u);
      const double lo = extra < nv[i] ? extra : nv[i];
      my_best = lo < my_best ? lo : my_best;
      // keep the slots' live ranges apart (otherwise every slot's LDS reads are hoisted to the top: +60 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!REPLAY) {
      if (any_slow) *bail = 1;  // seen by everyone after the reduction barrier below
    } else if (REPLAY && __any(any_slow)) {  // wave-uniform: some lane has a negative emission cost on an entry slot
      slow_taken = 1;
      my_best = kInf;
#pragma unroll
      for (int i = 0; i < SPT; i++) {
        const uint32_t p = tid + i * NT;
        const uint32_t inf = info[i];
        if (pk[i] & 0x80000000u) {
          const bool pos1 = inf & kSlotPos1;
          const double am = am_l[p], am_b = pos1 ? am_l[p - 1] : am;
          const double wp = (inf & kSlotSilWord) ? 0.0 : wp_word;
          const bool sil_state = inf & kSlotSilState;
          const double t_b = (pos1 && !(inf & kSlotFirstSil)) ? ts : tf;
          const Merge mg = replay_boundary(a.net.word_end_slot, a.net.n_words, sc, bk, p, pos1, !(inf & kSlotEnd), true, wp, t_b, am_b, am, tf,
                                           sil_state ? tf : tl, bkp_new);
          nv[i] = mg.score; pk[i] = mg.bkp;
        }
        double extra = kInf;
        if (p < P && (inf & kSlotSingle)) {
          const double am = am_l[p];
          const double wp = (inf & kSlotSilWord) ? 0.0 : wp_word;
          const double t_d = (inf & kSlotFirstSil) ? tf : ts;
          if (am >= 0.0) extra = ((m_we + wp) + t_d) + am;
          else extra = replay_boundary(a.net.word_end_slot, a.net.n_words, sc, bk, p, false, false, false, wp, t_d, am, am, tf, tf, bkp_new).score;
        }
        const double lo = extra < nv[i] ? extra : nv[i];
        my_best = lo < my_best ? lo : my_best;
      }
    }
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      const uint32_t p = tid + i * NT;
      pk[i] &= 0xFFFFu;
      if (p < P && (info[i] & kSlotEnd) && (nv[i] < my_we || (nv[i] == my_we && p < my_we_idx))) { my_we = nv[i]; my_we_idx = p; }
    }
    if (tid < 4) ef_nxt[tid] = 0xFFFFFFFFu;

    // ---- B: block reductions -----------------------------------------------------------------------
    my_best = wave_min(my_best);
    wave_min_idx(my_we, my_we_idx);
    if (lane == 0) { red_best[wave] = my_best; red_we[wave] = my_we; red_idx[wave] = my_we_idx; }
    __syncthreads();  // also: every read of sc/bk of frame t-1 is done
    if (!REPLAY && *bail) {  // workgroup-uniform
      if (tid == 0) { atomicOr(&a.out_flags[u], kFlagReplay); a.out_count[u] = 0; }
      return true;
    }
    double best = red_best[0], we = red_we[0];
    uint32_t we_idx = red_idx[0];
#pragma unroll
    for (uint32_t w = 1; w < kWavesPerWg; w++) {
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
    // a word end other than we_idx can only tie the boundary candidate of the minimum if it rounds to
    // the same sum: its score is within a few ulps of the minimum (cheap, safe pre-filter)
    const double near = m_we + (fabs(m_we) + fabs(wp_word) + fabs(tf) + fabs(ts) + 1.0) * 1e-9;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
      const uint32_t p = tid + i * NT;
      double v = nv[i];
      if (v > limit) v = kInf;  // :194-196
      sc[p] = v;
      bk[p] = (uint16_t)pk[i];
      am_l[p] = am_consume[i];  // next frame's emission costs (gathered two frames ago)
      if ((info[i] & kSlotEnd) && we_alive && v <= near) {
        if (p == we_idx) {  // first minimal surviving word end -> traceback[t] (:199-205)
          a.tb_score[tb0 + t] = v; a.tb_word[tb0 + t] = (uint16_t)p; a.tb_bkp[tb0 + t] = (uint16_t)pk[i];  // slot now, word after the loop
        }
        // first slot (in index order) whose boundary candidate equals the minimum's, per (wp, tdp) class
        if (v + 0.0 + tf == m_we + 0.0 + tf) atomicMin(&ef_nxt[0], p);
        if (v + 0.0 + ts == m_we + 0.0 + ts) atomicMin(&ef_nxt[1], p);
        if (v + wp_word + tf == m_we + wp_word + tf) atomicMin(&ef_nxt[2], p);
        if (v + wp_word + ts == m_we + wp_word + ts) atomicMin(&ef_nxt[3], p);
      }
    }
    if (!we_alive && tid == 0) { a.tb_score[tb0 + t] = kInf; a.tb_word[tb0 + t] = 0xFFFFu; a.tb_bkp[tb0 + t] = 0; }
    __syncthreads();
    return false;
  };

  for (uint32_t t = 1; t <= T; t += 2) {
    if (frame(t, am_s1, am_s0)) return;                 // odd frame: refill set 1 (frame t+2), mirror set 0 (frame t+1)
    if (t + 1 <= T && frame(t + 1, am_s0, am_s1)) return;
  }

  // ---- traceback (Recognizer.cpp:222-231) -------------------------------------------------------------
  __threadfence();
  __syncthreads();
  // entries 1..T hold the winning word-end SLOT (0xFFFF: no surviving word end -> word 0, :118,191): map to words
  bool bad = false;
  for (uint32_t t = 1 + tid; t <= T; t += NT) {
    const uint32_t sl = __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint16_t w = 0;
    if (sl != 0xFFFFu) { if (sl < P) w = (uint16_t)a.net.slot_word[sl]; else bad = true; }
    a.tb_word[tb0 + t] = w;
  }
  if (bad) atomicOr(&a.out_flags[u], kFlagCorrupt);
  __threadfence();
  __syncthreads();
  if (slow_taken) atomicOr(&a.out_flags[u], kFlagSlowPath);
  if (tid == 0) {  // guarded walk: traceback.h
    const uint32_t n = walk_traceback(
        T, a.net.silence_word, a.net.n_words,
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_bkp[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        a.out_words + f0, T);
    if (n == kTbCorrupt) atomicOr(&a.out_flags[u], kFlagCorrupt);
    a.out_count[u] = n == kTbCorrupt ? 0u : n;
  }
}

// ---- the same search for lexicons whose hypothesis arrays do not fit the LDS (more than 8192 slots) -----------------------
// Scores and back pointers of the two frames in flight live in a per-workgroup global workspace (20 bytes per slot: L2
// resident), emission costs are gathered where they are used, and a thread walks its slots in a loop instead of keeping them in
// registers.  Phases, merge order and the exactness argument are decode_kernel's (REPLAY = true: the sequential boundary
// replay is inline, per lane); slot ids stay 16 bit in the traceback, so P <= 65534.  A capacity path: measured 12.3 ms for
// 256 utterances at 9001 slots against 5.3 ms at 8000 slots in the type-sorted LDS kernel (2 x per slot).
template <int NT>
__global__ __launch_bounds__(NT) void decode_big_kernel(DecodeArgs a, unsigned char* ws_all, size_t ws_stride) {
  constexpr uint32_t kWavesPerWg = NT / 64;
  __shared__ double red_best[kWavesPerWg], red_we[kWavesPerWg];
  __shared__ uint32_t red_idx[kWavesPerWg], e_first[8];
  const uint32_t P = a.net.n_slots;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x;
  if (a.only_flagged && !(a.out_flags[u] & kFlagReplay)) return;  // workgroup-uniform: replay of the word-per-lane kernel, nothing to redo
  const uint64_t f0 = a.frame_off[u];
  const uint32_t T = (uint32_t)(a.frame_off[u + 1] - f0);
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint64_t tb0 = f0 + u;
  const double tl = a.net.tdp_loop, tf = a.net.tdp_forward, ts = a.net.tdp_skip;
  const double wp_word = a.word_penalty, thr = a.am_threshold;
  unsigned char* ws = ws_all + (size_t)blockIdx.x * ws_stride;
  double* scb[2] = {reinterpret_cast<double*>(ws), reinterpret_cast<double*>(ws) + P};
  uint16_t* bkb[2] = {reinterpret_cast<uint16_t*>(ws + (size_t)P * 16), reinterpret_cast<uint16_t*>(ws + (size_t)P * 16) + P};

  for (uint32_t p = tid; p < P; p += NT) { scb[0][p] = kInf; bkb[0][p] = 0; }
  if (tid < 8) e_first[tid] = 0xFFFFFFFFu;
  __syncthreads();
  const bool init_is_end = a.net.slot_info[0] & kSlotEnd;
  double m_we = init_is_end ? 0.0 : kInf;
  if (tid == 0) {
    scb[0][0] = 0.0;  // initial hypothesis: word 0, position 0, score 0 (Recognizer.cpp:120)
    a.tb_score[tb0] = 0.0; a.tb_word[tb0] = 0; a.tb_bkp[tb0] = 0;
  }
  if (tid < 4 && init_is_end) e_first[4 + tid] = 0;
  __syncthreads();
  uint32_t slow_taken = 0;

  for (uint32_t t = 1; t <= T; t++) {
    const uint32_t* ef_cur = e_first + 4 * (t & 1);
    uint32_t* ef_nxt = e_first + 4 * ((t + 1) & 1);
    const uint32_t bkp_new = (t - 1) & 0xFFFFu;
    const double* row = row0 + (uint64_t)(t - 1) * a.ld;
    const double* sc = scb[(t - 1) & 1];
    const uint16_t* bk = bkb[(t - 1) & 1];
    double* scn = scb[t & 1];
    uint16_t* bkn = bkb[t & 1];
    // ---- A: the new hypotheses, unpruned, into the other buffer ---------------------------------------------------
    double my_best = kInf, my_we = kInf;
    uint32_t my_we_idx = 0xFFFFFFFFu;
    for (uint32_t p = tid; p < P; p += NT) {
      const uint32_t inf = a.net.slot_info[p];
      const double am = row[inf & 0xFFFFu];
      const bool pos0 = inf & kSlotPos0, pos1 = inf & kSlotPos1, entry = pos0 || pos1;
      const bool sil_state = inf & kSlotSilState;
      const double t_loop = sil_state ? tf : tl, t_skip = sil_state ? tf : ts;
      const uint32_t p1 = p - (pos0 ? 0u : 1u), p2 = p - (entry ? 0u : 2u);
      const double c_skip = entry ? kInf : sc[p2] + t_skip;
      const double c_fwd = pos0 ? kInf : sc[p1] + tf;
      const double c_loop = (inf & kSlotEnd) ? kInf : sc[p] + t_loop;  // word ends do not expand in-word (:131)
      const double am_b = pos1 ? row[a.net.slot_info[p1] & 0xFFFFu] : am;  // emission of the word's position 0 (:136,148-151)
      const double wp = (inf & kSlotSilWord) ? 0.0 : wp_word;
      const bool b_skip = pos1 && !(inf & kSlotFirstSil);
      const double t_b = b_skip ? ts : tf;
      const uint32_t cls = ((inf & kSlotSilWord) ? 0u : 2u) + (b_skip ? 1u : 0u);
      Merge mg;
      if (entry && (am_b < 0.0 || am < 0.0)) {  // the collapsed boundary candidate is only exact while the pre-AM early-out is inert
        mg = replay_boundary(a.net.word_end_slot, a.net.n_words, sc, bk, p, pos1, !(inf & kSlotEnd), true, wp, t_b, am_b, am, tf, t_loop, bkp_new);
        slow_taken = 1;
      } else {
        const double c_b = entry ? (m_we + wp) + t_b : kInf;  // cur_hyp->score + word_penalty + tdp, :140
        const bool b_first = ef_cur[cls] < p1;                // is that source visited before the in-word ones?
        mg.offer(b_first ? c_b : kInf, am_b, bkp_new);
        mg.offer(c_skip, am, bk[p2]);
        mg.offer(c_fwd, am, bk[p1]);
        mg.offer(c_loop, am, bk[p]);
        mg.offer(b_first ? kInf : c_b, am_b, bkp_new);
      }
      double lo = mg.score;
      if (inf & kSlotSingle) {  // one-position word: its dead position-1 slot still feeds best_score (:139,155)
        const double t_d = (inf & kSlotFirstSil) ? tf : ts;
        double extra;
        if (am >= 0.0) extra = ((m_we + wp) + t_d) + am;
        else { extra = replay_boundary(a.net.word_end_slot, a.net.n_words, sc, bk, p, false, false, false, wp, t_d, am, am, tf, tf, bkp_new).score; slow_taken = 1; }
        lo = extra < lo ? extra : lo;
      }
      my_best = lo < my_best ? lo : my_best;
      scn[p] = mg.score;
      bkn[p] = (uint16_t)mg.bkp;
      if ((inf & kSlotEnd) && (mg.score < my_we || (mg.score == my_we && p < my_we_idx))) { my_we = mg.score; my_we_idx = p; }
    }
    if (tid < 4) ef_nxt[tid] = 0xFFFFFFFFu;
    // ---- B ---------------------------------------------------------------------------------------------------------
    my_best = wave_min(my_best);
    wave_min_idx(my_we, my_we_idx);
    if (lane == 0) { red_best[wave] = my_best; red_we[wave] = my_we; red_idx[wave] = my_we_idx; }
    __syncthreads();
    double best = red_best[0], we = red_we[0];
    uint32_t we_idx = red_idx[0];
#pragma unroll
    for (uint32_t w = 1; w < kWavesPerWg; w++) {
      const double ob = red_best[w];
      best = ob < best ? ob : best;
      const double ow = red_we[w];
      const uint32_t oi = red_idx[w];
      if (ow < we || (ow == we && oi < we_idx)) { we = ow; we_idx = oi; }
    }
    // ---- C: prune, traceback, publish the word-end minimum -----------------------------------------------------------
    const double limit = best + thr;
    const bool we_alive = !(we > limit) && we != kInf;
    m_we = we_alive ? we : kInf;
    const double near = m_we + (fabs(m_we) + fabs(wp_word) + fabs(tf) + fabs(ts) + 1.0) * 1e-9;
    for (uint32_t p = tid; p < P; p += NT) {  // (a thread prunes the slots it wrote)
      const double v = scn[p];
      if (v > limit) { scn[p] = kInf; continue; }  // :194-196
      if ((a.net.slot_info[p] & kSlotEnd) && we_alive && v <= near) {
        if (p == we_idx) { a.tb_score[tb0 + t] = v; a.tb_word[tb0 + t] = (uint16_t)p; a.tb_bkp[tb0 + t] = bkn[p]; }
        if (v + 0.0 + tf == m_we + 0.0 + tf) atomicMin(&ef_nxt[0], p);
        if (v + 0.0 + ts == m_we + 0.0 + ts) atomicMin(&ef_nxt[1], p);
        if (v + wp_word + tf == m_we + wp_word + tf) atomicMin(&ef_nxt[2], p);
        if (v + wp_word + ts == m_we + wp_word + ts) atomicMin(&ef_nxt[3], p);
      }
    }
    if (!we_alive && tid == 0) { a.tb_score[tb0 + t] = kInf; a.tb_word[tb0 + t] = 0xFFFFu; a.tb_bkp[tb0 + t] = 0; }
    __syncthreads();  // workgroup-scope release/acquire: the pruned scores are visible to every thread of the next frame
  }

  // ---- traceback (Recognizer.cpp:222-231) -------------------------------------------------------------
  __threadfence();
  __syncthreads();
  bool bad = false;
  for (uint32_t t = 1 + tid; t <= T; t += NT) {
    const uint32_t sl = __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint16_t w = 0;
    if (sl != 0xFFFFu) { if (sl < P) w = (uint16_t)a.net.slot_word[sl]; else bad = true; }
    a.tb_word[tb0 + t] = w;
  }
  if (bad) atomicOr(&a.out_flags[u], kFlagCorrupt);
  __threadfence();
  __syncthreads();
  if (slow_taken) atomicOr(&a.out_flags[u], kFlagSlowPath);
  if (tid == 0) {  // guarded walk: traceback.h
    const uint32_t n = walk_traceback(
        T, a.net.silence_word, a.net.n_words,
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_word[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        [&](uint32_t t) -> uint32_t { return __hip_atomic_load(&a.tb_bkp[tb0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
        a.out_words + f0, T);
    if (n == kTbCorrupt) atomicOr(&a.out_flags[u], kFlagCorrupt);
    a.out_count[u] = n == kTbCorrupt ? 0u : n;
  }
}

uint32_t decode_big_max_slots() { return 65534; }  // slot ids are 16 bit in the traceback, 0xFFFF = no surviving word end
size_t decode_big_workspace(uint32_t n_slots) { return (((size_t)n_slots * 20) + 255) & ~(size_t)255; }  // per utterance in flight
hipError_t launch_decode_big(const DecodeArgs& a, unsigned char* ws, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  if (a.net.n_slots > decode_big_max_slots() || !ws) return hipErrorInvalidValue;
  hipLaunchKernelGGL((decode_big_kernel<1024>), dim3(a.n_utts), dim3(1024), 0, stream, a, ws, decode_big_workspace(a.net.n_slots));
  return hipGetLastError();
}

uint32_t decode_max_slots() { return 8192; }

static size_t decode_smem(uint32_t PP) { return (size_t)PP * 16 + 16 * 8 * 2 + 16 * 4 + 12 * 4 + (size_t)PP * 2 + 16; }

hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const uint32_t P = a.net.n_slots;
  const dim3 grid(a.n_utts);
  // a fast variant first (viterbi_words.hip for short-word lexica, else viterbi_fast.hip); then the replay variant, whose workgroups exit at once unless the fast
  // one flagged their utterance (negative emission cost)
  if (!a.force_general) {
    hipError_t e = (!a.force_slots && decode_words_applies(a)) ? launch_decode_words(a, stream) : launch_decode_fast(a, stream);
    if (e != hipSuccess) return e;
  }
#define SR_LAUNCH(NT, SPT)                                                                                   \
  do {                                                                                                       \
    const size_t smem = decode_smem((NT) * (SPT));                                                           \
    hipError_t e = hipFuncSetAttribute((const void*)decode_kernel<NT, SPT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((decode_kernel<NT, SPT, true>), grid, dim3(NT), smem, stream, a);                     \
    return hipGetLastError();                                                                                \
  } while (0)
  if (P <= 64) SR_LAUNCH(64, 1);
  if (P <= 256) SR_LAUNCH(64, 4);
  if (P <= 1024) SR_LAUNCH(256, 4);
  if (P <= 2048) SR_LAUNCH(256, 8);
  if (P <= 4096) SR_LAUNCH(512, 8);
  if (P <= 8192) SR_LAUNCH(1024, 8);
#undef SR_LAUNCH
  return hipErrorInvalidValue;
}

}  // namespace srgpu

// ---- the walk alone (sr_traceback_corpus): traceback arrays supplied by the caller, one thread per utterance ------------
namespace srgpu {
__global__ void traceback_kernel(const uint64_t* frame_off, uint32_t n_utts, const uint16_t* tb_word, const uint16_t* tb_bkp,
                                 uint32_t silence_word, uint32_t n_words, uint32_t* out_words, uint32_t* out_count,
                                 uint32_t* out_flags) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_utts) return;
  const uint64_t f0 = frame_off[u], tb0 = f0 + u;
  const uint32_t T = (uint32_t)(frame_off[u + 1] - f0);
  const uint32_t n = walk_traceback(
      T, silence_word, n_words, [&](uint32_t t) -> uint32_t { return tb_word[tb0 + t]; },
      [&](uint32_t t) -> uint32_t { return tb_bkp[tb0 + t]; }, out_words + f0, T);
  out_flags[u] = n == kTbCorrupt ? kFlagCorrupt : 0u;
  out_count[u] = n == kTbCorrupt ? 0u : n;
}

hipError_t launch_traceback(const uint64_t* frame_off, uint32_t n_utts, const uint16_t* tb_word, const uint16_t* tb_bkp,
                            uint32_t silence_word, uint32_t n_words, uint32_t* out_words, uint32_t* out_count,
                            uint32_t* out_flags, hipStream_t stream) {
  if (n_utts == 0) return hipSuccess;
  hipLaunchKernelGGL(traceback_kernel, dim3((n_utts + 63) / 64), dim3(64), 0, stream, frame_off, n_utts, tb_word, tb_bkp,
                     silence_word, n_words, out_words, out_count, out_flags);
  return hipGetLastError();
}
}  // namespace srgpu
