// viterbi_align.hip -- forced alignment of an utterance against a linear reference automaton:
//   align_full_kernel    Aligner::align_sequence_full   (sietill/Alignment.cpp:50-144)
//   align_pruned_kernel  Aligner::align_sequence_pruned (sietill/Alignment.cpp:149-288)
//
// One workgroup per utterance; trellis positions are strided over the threads, path costs live in
// LDS (ping-pong buffers), the emission row of the frame is gathered from the dense score table
// (8 B per touched position) and one byte of back-pointer per (frame, position) goes to HBM; the
// back-trace runs on the device at the end.  Algorithmic bytes per frame: (8 + 1) * N.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

static constexpr double kInf = __builtin_huge_val();
static constexpr int kAlignThreads = 256;

__device__ inline double tdp_score(uint32_t to, int jump, uint32_t sil, double tl, double tf, double ts) {
  if (to == sil) return tf;  // TdpModel.cpp:20-22
  return jump == 0 ? tl : (jump == 1 ? tf : ts);
}

__device__ inline double shfl_xor_f64a(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// LDS layout: cost[2][N] f64, ref[N] u16
__global__ __launch_bounds__(kAlignThreads) void align_full_kernel(AlignArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x, tid = threadIdx.x;
  const uint64_t f0 = a.frame_off[u];
  const int T = (int)(a.frame_off[u + 1] - f0);
  const int N = (int)(a.aut_off[u + 1] - a.aut_off[u]);
  const uint16_t* ref_g = a.automata + a.aut_off[u];
  double* cost = reinterpret_cast<double*>(smem);
  uint16_t* ref = reinterpret_cast<uint16_t*>(cost + 2 * (size_t)N);
  uint8_t* bp = a.backptr + a.bp_off[u];  // [T][N]
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint32_t sil = a.silence_state;
  const double tl = a.tdp_loop, tf = a.tdp_forward, ts = a.tdp_skip;

  for (int s = tid; s < N; s += kAlignThreads) {
    ref[s] = ref_g[s];
    cost[s] = kInf;      // previous_costs / current_costs start at +inf (Alignment.cpp:62-63)
    cost[N + s] = kInf;
  }
  __syncthreads();
  if (tid == 0) cost[0] = row0[ref[0]];  // :77
  __syncthreads();

  for (int t = 1; t < T; t++) {
    const double* prev = cost + (size_t)((t - 1) & 1) * N;
    double* cur = cost + (size_t)(t & 1) * N;
    const double* row = row0 + (uint64_t)t * a.ld;
    // reachable window of the 0-1-2 topology (:73-74, :82, :87)
    const int lo_raw = N - 1 - 2 * (T - 1 - t);
    const int lo = lo_raw > 0 ? lo_raw : 0;
    const int hi = (N - 1 < 2 * t) ? N - 1 : 2 * t;
    for (int s = lo + (int)tid; s <= hi; s += kAlignThreads) {
      const double local = row[ref[s]];
      double best = prev[s] + tdp_score(ref[s], 0, sil, tl, tf, ts);  // loop, keyed on the state itself (:95)
      int taken = 0;
      if (s > 0) {
        const double fw = prev[s - 1] + tdp_score(ref[s - 1], 1, sil, tl, tf, ts);  // keyed on the SOURCE (:100)
        if (fw < best) { best = fw; taken = 1; }
      }
      if (s > 1) {
        const double sk = prev[s - 2] + tdp_score(ref[s - 2], 2, sil, tl, tf, ts);
        if (sk < best) { best = sk; taken = 2; }
      }
      cur[s] = local + best;  // :115
      bp[(size_t)t * N + s] = (uint8_t)taken;
    }
    __syncthreads();
  }

  __threadfence();
  __syncthreads();
  if (tid == 0) {
    // with T == 1 nothing was ever written to current_costs: the reference returns +inf (:143)
    a.out_cost[u] = (T >= 2) ? cost[(size_t)((T - 1) & 1) * N + (N - 1)] : kInf;
    uint16_t* out = a.out_states + f0;
    int si = N - 1;
    for (int t = T - 1; t >= 0; t--) {  // :129-138
      out[t] = ref[si];
      if (t > 0) si -= (int)__hip_atomic_load(&bp[(size_t)t * N + si], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LDS layout: cost[2][N] f64, ref[N] u16, alive[2][N] u8, red[4] f64
__global__ __launch_bounds__(kAlignThreads) void align_pruned_kernel(AlignArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t u = a.utt_order ? a.utt_order[a.utt_first + blockIdx.x] : a.utt_first + blockIdx.x, tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63;
  const uint64_t f0 = a.frame_off[u];
  const int T = (int)(a.frame_off[u + 1] - f0);
  const int N = (int)(a.aut_off[u + 1] - a.aut_off[u]);
  const uint16_t* ref_g = a.automata + a.aut_off[u];
  double* cost = reinterpret_cast<double*>(smem);
  double* red = cost + 2 * (size_t)N;
  uint16_t* ref = reinterpret_cast<uint16_t*>(red + 4);
  uint8_t* alive = reinterpret_cast<uint8_t*>(ref + N);
  uint8_t* bp = a.backptr + a.bp_off[u];
  const double* row0 = a.scores + (f0 - a.frame_base) * a.ld;
  const uint32_t sil = a.silence_state;
  const double tl = a.tdp_loop, tf = a.tdp_forward, ts = a.tdp_skip, thr = a.pruning_threshold;

  for (int s = tid; s < N; s += kAlignThreads) {
    ref[s] = ref_g[s];
    alive[s] = 0;
    alive[N + s] = 0;
  }
  __syncthreads();
  if (tid == 0) { cost[0] = row0[ref[0]]; alive[0] = 1; }  // initial node (:158-160)
  __syncthreads();

  for (int t = 1; t < T; t++) {
    const double* prev = cost + (size_t)((t - 1) & 1) * N;
    const uint8_t* pal = alive + (size_t)((t - 1) & 1) * N;
    double* cur = cost + (size_t)(t & 1) * N;
    uint8_t* cal = alive + (size_t)(t & 1) * N;
    const double* row = row0 + (uint64_t)t * a.ld;
    double my_best = kInf;
    for (int q = tid; q < N; q += kAlignThreads) {
      // sources arrive in ascending position: q-2 (skip), q-1 (forward), q (loop); the first creates the
      // node, a later one replaces it only when strictly better (:200-205)
      const double am = row[ref[q]];
      bool have = false;
      double c = kInf;
      int taken = 3;
      for (int jump = 2; jump >= 0; jump--) {
        const int p = q - jump;
        if (p < 0 || !pal[p]) continue;
        double n = prev[p];
        n += tdp_score(ref[q], jump, sil, tl, tf, ts);  // keyed on the DESTINATION (:191)
        n += am;
        if (!have) { have = true; c = n; taken = jump; }
        else if (c > n) { c = n; taken = jump; }
      }
      cur[q] = c;
      cal[q] = have ? 1 : 0;
      bp[(size_t)t * N + q] = (uint8_t)taken;
      if (have && my_best > c) my_best = c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double o = shfl_xor_f64a(my_best, m);
      my_best = o < my_best ? o : my_best;
    }
    if (lane == 0) red[wave] = my_best;
    __syncthreads();
    double best = red[0];
    for (int w = 1; w < kAlignThreads / 64; w++) best = red[w] < best ? red[w] : best;
    const double ub = best + thr;  // :226
    for (int q = tid; q < N; q += kAlignThreads)
      if (cal[q] && cur[q] > ub) cal[q] = 0;
    __syncthreads();
  }

  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const uint8_t* lal = alive + (size_t)((T - 1) & 1) * N;
    const double* lc = cost + (size_t)((T - 1) & 1) * N;
    int hi = 0;
    for (int q = 0; q < N; q++) if (lal[q]) hi = q;  // highest position reached (:244-251)
    a.out_cost[u] = lc[hi];
    uint16_t* out = a.out_states + f0;
    int p = hi;
    for (int t = T - 1; t > 0; t--) {  // :258-270
      out[t] = ref[p];
      p -= (int)__hip_atomic_load(&bp[(size_t)t * N + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[0] = ref[0];  // :273-276
  }
}

__global__ void path_scores_kernel(const double* scores, uint32_t ld, uint64_t frame_base, uint64_t f0, uint64_t f1,
                                   const uint16_t* states, double* out) {
  const uint64_t f = f0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < f1) out[f] = scores[(f - frame_base) * ld + states[f]];
}

hipError_t launch_path_scores(const double* scores, uint32_t ld, uint64_t frame_base, uint64_t f0, uint64_t f1,
                              const uint16_t* states, double* out, hipStream_t stream) {
  if (f1 <= f0) return hipSuccess;
  hipLaunchKernelGGL(path_scores_kernel, dim3((unsigned)((f1 - f0 + 255) / 256)), dim3(256), 0, stream, scores, ld, frame_base,
                     f0, f1, states, out);
  return hipGetLastError();
}

uint32_t align_max_positions() { return 8192; }

hipError_t launch_align_full(const AlignArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const size_t smem = (size_t)a.max_positions * (2 * 8 + 2) + 16;
  hipError_t e = hipFuncSetAttribute((const void*)align_full_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(align_full_kernel, dim3(a.n_utts), dim3(kAlignThreads), smem, stream, a);
  return hipGetLastError();
}

hipError_t launch_align_pruned(const AlignArgs& a, hipStream_t stream) {
  if (a.n_utts == 0) return hipSuccess;
  const size_t smem = (size_t)a.max_positions * (2 * 8 + 2 + 2) + 4 * 8 + 16;
  hipError_t e = hipFuncSetAttribute((const void*)align_pruned_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(align_pruned_kernel, dim3(a.n_utts), dim3(kAlignThreads), smem, stream, a);
  return hipGetLastError();
}

}  // namespace srgpu
