// gmm_exact.hip -- direct-form diagonal-GMM scoring that replays the reference's floating-point
// operation order, so max-approx scores are BIT-IDENTICAL to MixtureModel::score on x86
// (sietill/Mixtures.cpp:645-690 density_score_sse, :696-713 min_score): two partial sums over
// even / odd dimensions (the two SSE lanes), l0 + l1, a scalar tail for odd dim, norm + dist/2,
// then - logw; no fused multiply-add anywhere (the reference is built -msse3 only,
// sietill/Makefile:22).  Sum mode (:719-728) uses the same density scores but the device exp/log,
// so it is accurate to an ulp or two rather than bit-identical.
//
// One thread owns one frame (features converted to double once, kept in registers for dim 39/25 or
// in LDS otherwise); the state loop is wave-uniform, so model rows arrive through the scalar cache
// and every vector instruction is FP64 VALU work: 4*dim unfused ops per (frame, density).  FP64
// vector peak equals FP64 matrix peak on MI355X, which puts this kernel at half the MFMA kernel's
// ceiling -- it is the on-device parity oracle and the tie-breaker for the MFMA path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

#pragma clang fp contract(off)

static constexpr int kExactThreads = 256;

template <int DS, bool SUM>
__global__ __launch_bounds__(kExactThreads) void gmm_exact_kernel(
    const float* __restrict__ feats, uint64_t n_frames, uint32_t dim, uint32_t n_states,
    const uint32_t* __restrict__ dens_off, const double* __restrict__ means, const double* __restrict__ inv_vars,
    const double* __restrict__ norm, const double* __restrict__ logw, double* __restrict__ out, uint32_t ld,
    uint32_t states_per_split, GmmExactList L) {
  extern __shared__ float xs[];  // generic-dim path: [dim][kExactThreads]
  const uint32_t D = DS ? DS : dim;
  // dense: block x = 256 consecutive frames, block y = a state range.  listed (L.states != null): block x = up to 256
  // frames of ONE utterance, scored for the states of that utterance's list only (the forced aligner's automaton)
  uint64_t f, f_out;
  bool valid;
  uint32_t s0, s1;
  if (L.states) {
    const uint32_t b = L.blk_first + blockIdx.x;
    f = L.blk_frame0[b] + threadIdx.x;
    valid = threadIdx.x < L.blk_frames[b];
    f_out = f - L.frame_base;
    s0 = L.list_off[L.blk_list[b]];
    s1 = L.list_off[L.blk_list[b] + 1];
    if (!valid) { f = L.blk_frame0[b]; f_out = 0; }
  } else {
    f = (uint64_t)blockIdx.x * kExactThreads + threadIdx.x;
    valid = f < n_frames;
    if (!valid) f = 0;
    f_out = f;
    s0 = blockIdx.y * states_per_split;
    s1 = (s0 + states_per_split < n_states) ? s0 + states_per_split : n_states;
  }
  const float* xrow = feats + f * D;
  double xr[DS ? DS : 1];
  if (DS) {
#pragma unroll
    for (int d = 0; d < DS; d++) xr[d] = (double)xrow[d];
  } else {
    for (uint32_t d = 0; d < D; d++) xs[d * kExactThreads + threadIdx.x] = xrow[d];
  }
  const uint32_t D2 = D - (D & 1u);

  for (uint32_t si = s0; si < s1; si++) {
    const uint32_t s = L.states ? L.states[si] : si;
    const uint32_t c0 = dens_off[s], c1 = dens_off[s + 1];
    double best = SUM ? 0.0 : 1e10;  // Mixtures.cpp:699 / :721
    for (uint32_t c = c0; c < c1; c++) {
      const double* mu = means + (uint64_t)c * D;
      const double* iv = inv_vars + (uint64_t)c * D;
      double l0 = 0.0, l1 = 0.0;
      if (DS) {
#pragma unroll
        for (int d = 0; d < (DS & ~1); d += 2) {
          double p = xr[d] - mu[d];
          p = p * p;
          p = p * iv[d];
          l0 = l0 + p;
          double q = xr[d + 1] - mu[d + 1];
          q = q * q;
          q = q * iv[d + 1];
          l1 = l1 + q;
        }
      } else {
        for (uint32_t d = 0; d < D2; d += 2) {
          double p = (double)xs[d * kExactThreads + threadIdx.x] - mu[d];
          p = p * p;
          p = p * iv[d];
          l0 = l0 + p;
          double q = (double)xs[(d + 1) * kExactThreads + threadIdx.x] - mu[d + 1];
          q = q * q;
          q = q * iv[d + 1];
          l1 = l1 + q;
        }
      }
      double dist = l0 + l1;
      if (D & 1u) {
        const double xl = DS ? xr[DS ? DS - 1 : 0] : (double)xs[(D - 1) * kExactThreads + threadIdx.x];
        const double t = xl - mu[D - 1];
        dist += t * t * iv[D - 1];
      }
      double score = norm[c] + dist / 2;
      score -= logw[c];
      if (SUM) best += exp(-1 * score);
      else if (score < best) best = score;
    }
    if (valid) out[f_out * ld + s] = SUM ? -1 * log(best) : best;
  }
}

template <int DS>
static hipError_t launch_d(const GmmExactArgs& a, bool sum, uint32_t n_splits, const GmmExactList& L, uint32_t n_blocks,
                           hipStream_t stream) {
  const dim3 grid(L.states ? n_blocks : (unsigned)((a.n_frames + kExactThreads - 1) / kExactThreads), L.states ? 1 : n_splits),
      block(kExactThreads);
  const size_t shmem = DS ? 0 : (size_t)a.dim * kExactThreads * sizeof(float);
  if (shmem > 160 * 1024) return hipErrorInvalidValue;  // (dim <= 160: srhost::model_shell refuses larger models)
  if (shmem > 48 * 1024) {  // the run-time-dimension path keeps a workgroup's frames in LDS: beyond the default dynamic size from dim 48
    hipError_t e = sum ? hipFuncSetAttribute((const void*)gmm_exact_kernel<DS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)
                       : hipFuncSetAttribute((const void*)gmm_exact_kernel<DS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
  }
  if (sum)
    hipLaunchKernelGGL((gmm_exact_kernel<DS, true>), grid, block, shmem, stream, a.feats, a.n_frames, a.dim, a.n_states,
                       a.dens_off, a.means, a.inv_vars, a.norm, a.logw, a.out, a.ld, a.states_per_split, L);
  else
    hipLaunchKernelGGL((gmm_exact_kernel<DS, false>), grid, block, shmem, stream, a.feats, a.n_frames, a.dim, a.n_states,
                       a.dens_off, a.means, a.inv_vars, a.norm, a.logw, a.out, a.ld, a.states_per_split, L);
  return hipGetLastError();
}

hipError_t launch_gmm_exact(const GmmExactArgs& a, bool sum, uint32_t n_splits, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  const GmmExactList none{};
  switch (a.dim) {
    case 39: return launch_d<39>(a, sum, n_splits, none, 0, stream);
    case 25: return launch_d<25>(a, sum, n_splits, none, 0, stream);
    default: return launch_d<0>(a, sum, n_splits, none, 0, stream);
  }
}

hipError_t launch_gmm_exact_listed(const GmmExactArgs& a, bool sum, const GmmExactList& L, uint32_t n_blocks, hipStream_t stream) {
  if (n_blocks == 0) return hipSuccess;
  switch (a.dim) {
    case 39: return launch_d<39>(a, sum, 1, L, n_blocks, stream);
    case 25: return launch_d<25>(a, sum, 1, L, n_blocks, stream);
    default: return launch_d<0>(a, sum, 1, L, n_blocks, stream);
  }
}

int gmm_exact_frames_per_block() { return kExactThreads; }

}  // namespace srgpu
