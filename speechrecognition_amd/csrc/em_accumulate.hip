// em_accumulate.hip -- EM statistics of an alignment on the GPU: MixtureModel::accumulate
// (sietill/Mixtures.cpp:278-372) after reset_accumulators (:235-247).  The step on the far side of the aligner
// in Trainer::train (Training.cpp:142-151,198-202).
//
// The reference adds frame after frame into per-density double accumulators; floating-point addition is not
// associative, so to stay BIT-IDENTICAL in the max-approx and first-pass modes the frames of every
// accumulator row are summed sequentially in frame order here too:
//   1. em_assign_kernel   one thread per frame: (frame, density, weight) "pairs" -- the arg-min density of the
//                         aligned mixture (min_score, :696-713, replaying density_score_sse's operation order),
//                         density 0 on the first pass, or every density with its membership exp(-score)/sum
//                         in soft mode (entries below 1e-8 are dropped like :334-336);
//   2. a STABLE radix sort of the pairs by mean index (and again by variance index: tied variances are summed
//      across the densities sharing them, in frame order);
//   3. em_bounds_kernel   first pair of every accumulator row (binary search per row), then
//      em_sum_kernel      one wave per row (a lane per dimension) walks the row's pairs in order:
//                         mean_acc += w*x, var_acc += (w*x)*x (starting at 1e-4, :243), weight += w.
// Parallelism is across rows (10^5 of them), never inside a row.  Soft mode uses the device exp, so its
// weights differ from glibc's by an ulp or two: tolerance 1e-12 there, bit-exact otherwise.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

#pragma clang fp contract(off)

static constexpr int kAssignThreads = 128;

__device__ inline double em_density_score(const float* x, const double* mu, const double* iv, double norm, double logw,
                                          uint32_t D) {
  double l0 = 0.0, l1 = 0.0;
  const uint32_t D2 = D - (D & 1u);
  for (uint32_t d = 0; d < D2; d += 2) {  // Mixtures.cpp:651-667: two partial sums (the SSE lanes)
    double p = (double)x[d] - mu[d];
    p = p * p;
    p = p * iv[d];
    l0 = l0 + p;
    double q = (double)x[d + 1] - mu[d + 1];
    q = q * q;
    q = q * iv[d + 1];
    l1 = l1 + q;
  }
  double dist = l0 + l1;
  if (D & 1u) {
    const double t = (double)x[D - 1] - mu[D - 1];
    dist += t * t * iv[D - 1];
  }
  double score = norm + dist / 2;
  score -= logw;
  return score;
}

__global__ __launch_bounds__(kAssignThreads) void em_assign_kernel(EmArgs a) {
  const uint64_t t = (uint64_t)blockIdx.x * kAssignThreads + threadIdx.x;
  if (t >= a.n_frames) return;
  const uint32_t D = a.dim;
  const float* x = a.feats + t * D;
  const uint32_t s = a.states[t], c0 = a.dens_off[s], c1 = a.dens_off[s + 1];
  const uint64_t p0 = a.pair_off[t];  // first pair of this frame; one pair per density in soft mode, else one
  if (a.first_pass || a.max_approx) {
    uint32_t arg = c0;
    if (!a.first_pass) {
      double best = 1e10;  // min_score: seed 1e10, index 0, strict <
      for (uint32_t c = c0; c < c1; c++) {
        const double sc = em_density_score(x, a.means + (uint64_t)c * D, a.inv_vars + (uint64_t)c * D, a.norm[c], a.logw[c], D);
        if (sc < best) { best = sc; arg = c; }
      }
    }
    const bool any = c1 > c0;  // an empty mixture contributes nothing
    a.pair_frame[p0] = (uint32_t)t;
    a.pair_w[p0] = 1.0;
    a.key_mean[p0] = any ? a.dens_mean[arg] : 0xFFFFFFFFu;
    a.key_var[p0] = any ? a.dens_var[arg] : 0xFFFFFFFFu;
    return;
  }
  double sum = 0.0;  // soft memberships: std::accumulate from 0.0 in density order (:320-321)
  for (uint32_t c = c0; c < c1; c++) {
    const double p = exp(-1 * em_density_score(x, a.means + (uint64_t)c * D, a.inv_vars + (uint64_t)c * D, a.norm[c], a.logw[c], D));
    a.pair_w[p0 + (c - c0)] = p;
    sum += p;
  }
  for (uint32_t c = c0; c < c1; c++) {
    const uint64_t i = p0 + (c - c0);
    const double p = a.pair_w[i] / sum;
    const bool keep = !(p < 1e-8);  // :334-336
    a.pair_frame[i] = (uint32_t)t;
    a.pair_w[i] = p;
    a.key_mean[i] = keep ? a.dens_mean[c] : 0xFFFFFFFFu;
    a.key_var[i] = keep ? a.dens_var[c] : 0xFFFFFFFFu;
  }
}

// out[t] = MixtureModel::score(frame t, states[t]) (Trainer::calc_am_score's summand, Training.cpp:605): one thread per
// frame, direct form in the reference's operation order -- the same bits as the dense table's entry
__global__ __launch_bounds__(kAssignThreads) void path_score_direct_kernel(EmArgs a, double* out) {
  const uint64_t t = (uint64_t)blockIdx.x * kAssignThreads + threadIdx.x;
  if (t >= a.n_frames) return;
  const uint32_t D = a.dim;
  const float* x = a.feats + t * D;
  const uint32_t s = a.states[t], c0 = a.dens_off[s], c1 = a.dens_off[s + 1];
  if (a.max_approx) {
    double best = 1e10;
    for (uint32_t c = c0; c < c1; c++) {
      const double sc = em_density_score(x, a.means + (uint64_t)c * D, a.inv_vars + (uint64_t)c * D, a.norm[c], a.logw[c], D);
      if (sc < best) best = sc;
    }
    out[t] = best;
  } else {
    double sum = 0.0;  // sum_score (Mixtures.cpp:719-728)
    for (uint32_t c = c0; c < c1; c++)
      sum += exp(-1 * em_density_score(x, a.means + (uint64_t)c * D, a.inv_vars + (uint64_t)c * D, a.norm[c], a.logw[c], D));
    out[t] = -1 * log(sum);
  }
}

hipError_t launch_path_scores_direct(const EmArgs& a, double* out, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  hipLaunchKernelGGL(path_score_direct_kernel, dim3((unsigned)((a.n_frames + kAssignThreads - 1) / kAssignThreads)),
                     dim3(kAssignThreads), 0, stream, a, out);
  return hipGetLastError();
}

// row_begin[r] = first pair whose key is >= r (r = 0 .. n_rows): one thread per row, a binary search each
__global__ __launch_bounds__(256) void em_bounds_kernel(const uint32_t* sorted_keys, uint64_t n_pairs, uint32_t n_rows,
                                                        uint32_t* row_begin) {
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row > n_rows) return;
  uint64_t lo = 0, hi = n_pairs;
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (sorted_keys[mid] < row) lo = mid + 1; else hi = mid; }
  row_begin[row] = (uint32_t)lo;
}

// One WAVE per accumulator row, a lane per dimension; SQUARE: variance statistics.  The additions of a row must stay in corpus
// order (bit-exactness), so a row is never split -- but a row that collects tens of thousands of frames (silence) is bound by
// memory round trips, not by its chain of additions: the wave fetches the row's pairs 64 at a time (ids, then weights and
// frames: one lane each, the next batch while this one is summed), hands them round through a wave-private LDS slot array (broadcast
// reads), and keeps 32 feature rows (156 contiguous bytes each) in flight.  (One thread per (row, dimension) walking the list by itself: 14 ms for the
// variance sums of the bench model against 4.5 with its loads staged 16 at a time and 1.x this way.)
template <bool SQUARE>
__global__ __launch_bounds__(256) void em_sum_kernel(EmArgs a, const uint32_t* row_begin, const uint32_t* sorted_pairs,
                                                     uint32_t n_rows, double* acc, double* weight) {
  const uint32_t D = a.dim, lane = threadIdx.x & 63u;
  const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (row >= n_rows) return;  // wave-uniform
  const uint32_t lo = row_begin[row], e = row_begin[row + 1];
  constexpr int kF = 32;  // feature rows in flight
  __shared__ ulonglong2 slots[4][64];
  ulonglong2* slot = slots[threadIdx.x >> 6];  // wave-private: (feature row offset, weight) of the batch's pairs
  for (uint32_t d0 = 0; d0 < D; d0 += 64) {  // (one pass for D <= 64)
    const uint32_t d = d0 + lane;
    const bool act = d < D;
    const float* col = a.feats + (act ? d : 0u);
    double sum = SQUARE ? 1e-4 : 0.0;  // reset_accumulators: variances start at minimal_variance_value_ (:167,243)
    double w = 0.0;
    // batch metadata of this lane: pair lo + lane of the current batch
    auto meta = [&](uint32_t i, double& pw, uint32_t& fr) {
      const uint32_t pr = sorted_pairs[(i + lane < e) ? i + lane : (e ? e - 1 : 0u)];
      pw = a.pair_w[pr];
      fr = a.pair_frame[pr];
    };
    double pw = 0.0, pw_n = 0.0;
    uint32_t fr = 0, fr_n = 0;
    if (lo < e) meta(lo, pw, fr);
    for (uint32_t i = lo; i < e; i += 64) {
      const uint32_t n = (e - i < 64u) ? e - i : 64u;
      if (i + 64 < e) meta(i + 64, pw_n, fr_n);  // the next batch's metadata travels while this batch is summed
      // this lane's pair to the wave's LDS slots: the loop below reads element j from ONE address (a broadcast read)
      // instead of three v_readlane + the offset multiplication per element
      // (wave-scope release / acquire around the exchange: the slots are written by one lane each and read by all, and the
      // next batch overwrites them -- without the fences only the in-order LDS pipe and the compiler's may-alias ordering
      // kept that correct, and the bit-exact statistics depend on it; they cost nothing)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // every lane's reads of the previous batch come before ...
      slot[lane] = make_ulonglong2((unsigned long long)fr * D, (unsigned long long)__double_as_longlong(pw));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // ... and every lane's store is visible to the reads below
      for (uint32_t j0 = 0; j0 < n; j0 += kF) {  // wave-uniform
        double yy[kF];
#pragma unroll
        for (int j = 0; j < kF; j++) yy[j] = (double)col[slot[(j0 + j < n) ? j0 + j : j0].x];
#pragma unroll
        for (int j = 0; j < kF; j++) {
          if (j0 + j < n) {  // wave-uniform
            const double p = __longlong_as_double((long long)slot[j0 + j].y);
            if (SQUARE) sum = sum + p * yy[j] * yy[j];  // scale_add_square: x + scale * y * y  (:56-64)
            else sum = sum + p * yy[j];                 // scale_add:        x + scale * y      (:46-54)
            w += p;
          }
        }
      }
      pw = pw_n; fr = fr_n;
    }
    if (act) acc[(uint64_t)row * D + d] = sum;
    if (d0 == 0 && lane == 0) weight[row] = w;
  }
}

size_t em_sort_temp_bytes(uint64_t n_pairs) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)n_pairs);
  return bytes;
}

hipError_t launch_em_accumulate(const EmArgs& a, void* sort_temp, size_t sort_temp_bytes, uint32_t* iota, uint32_t* keys_sorted,
                                uint32_t* pairs_sorted, uint32_t* row_begin, double* mean_acc, double* mean_w, double* var_acc, double* var_w,
                                hipStream_t stream) {
  if (a.n_frames == 0 || a.n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(em_assign_kernel, dim3((unsigned)((a.n_frames + kAssignThreads - 1) / kAssignThreads)), dim3(kAssignThreads), 0,
                     stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // LSD radix sort is stable: pairs of one row stay in generation (= frame, then density) order
  e = hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_temp_bytes, a.key_mean, keys_sorted, iota, pairs_sorted, (int)a.n_pairs, 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(em_bounds_kernel, dim3(a.n_mean / 256 + 1), dim3(256), 0, stream, keys_sorted, a.n_pairs, a.n_mean, row_begin);
  hipLaunchKernelGGL((em_sum_kernel<false>), dim3((a.n_mean + 3) / 4), dim3(256), 0, stream, a,
                     row_begin, pairs_sorted, a.n_mean, mean_acc, mean_w);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  e = hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_temp_bytes, a.key_var, keys_sorted, iota, pairs_sorted, (int)a.n_pairs, 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(em_bounds_kernel, dim3(a.n_var / 256 + 1), dim3(256), 0, stream, keys_sorted, a.n_pairs, a.n_var, row_begin);
  hipLaunchKernelGGL((em_sum_kernel<true>), dim3((a.n_var + 3) / 4), dim3(256), 0, stream, a,
                     row_begin, pairs_sorted, a.n_var, var_acc, var_w);
  return hipGetLastError();
}

}  // namespace srgpu
