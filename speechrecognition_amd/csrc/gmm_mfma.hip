// gmm_mfma.hip -- diagonal-GMM scoring as an FP64 MFMA contraction with a fused mixture epilogue.
//
// Replaces the reference's per-(frame,state) scoring loop MixtureModel::score -> min_score /
// sum_score -> density_score_sse (sietill/Mixtures.cpp:737-744, :696-728, :645-690) by one dense
// contraction over (frames x densities x 2*dim+1):
//
//   score(t,c) = norm_c - logw_c + 1/2 sum_d (x_td - mu_cd)^2 iv_cd
//              = sum_d [ (iv_cd/2) x_td^2 + (-mu_cd iv_cd) x_td ] + (norm_c - logw_c + 1/2 sum_d mu_cd^2 iv_cd)
//              = A[c][:] . B[:][t],   K = 2*dim+1  (padded to 4*KSTEPS)
//
// with A packed on the host in MFMA fragment order and B = [x^2, x, 1] built in registers from the
// float32 features (x^2 of a float is exact in double).  v_mfma_f64_16x16x4_f64: A = model rows,
// B = frames, D[row = density, col = frame] with row = (lane>>4) + 4*reg.  Model rows are laid out
// so that row r of a 16-row block is density (r>>2) of state-slot (r&3) of a 4-state group: the
// four accumulator registers of a lane are then four densities of ONE state for ONE frame and the
// mixture min (or sum of exp) needs no cross-lane traffic at all.
//
// Work split: workgroup = 4 waves x (NB*16) frames; each wave keeps its B fragments (NB*KSTEPS
// doubles per lane) in registers for the whole kernel and streams the model blocks of its
// state-range split through a double-buffered LDS stage filled by global_load_lds (16 B/lane).
// MFMA issue is the bound (FP64 matrix peak 78.6 TFLOP/s); bytes are negligible (64 flop/B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

typedef double v4d __attribute__((ext_vector_type(4)));

static constexpr int kWaves = 4;     // waves per workgroup
static constexpr int kStageBlocks = 2;  // model blocks per LDS stage

template <int KSTEPS, int NB, bool SUM>
__global__ __launch_bounds__(kWaves * 64, 2) void gmm_mfma_kernel(GmmMfmaArgs a) {
  constexpr int kBlockDoubles = KSTEPS * 64;             // one 16-row model block, fragment order
  constexpr int kStageDoubles = kStageBlocks * kBlockDoubles;
  constexpr int kChunksPerBlock = KSTEPS / 2;            // 1 KiB chunks (2 k-steps each)
  constexpr int kChunksPerStage = kStageBlocks * kChunksPerBlock;
  static_assert(KSTEPS % 2 == 0, "KSTEPS must be even (1 KiB LDS-DMA chunks)");
  __shared__ __attribute__((aligned(1024))) double lds[2 * kStageDoubles];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;   // k sub-index for A/B fragments; state slot of the result rows
  const int c = lane & 15;   // frame column inside a 16-frame block

  // XCD-aware tile mapping: workgroups with equal (id % 8) share an XCD (and its L2); give each
  // XCD whole state-range splits so co-resident workgroups stream the same model panel.
  uint32_t x, y;
  {
    const uint32_t id = blockIdx.x;
    if ((a.ny & 7u) == 0) {
      const uint32_t j = id >> 3;
      y = (id & 7u) + 8u * (j / a.nx);
      x = j % a.nx;
    } else {
      y = id / a.nx;
      x = id % a.nx;
    }
  }
  const uint32_t b0 = a.split_begin[y], b1 = a.split_begin[y + 1];
  const uint64_t frame0 = (uint64_t)x * (kWaves * NB * 16) + (uint64_t)wave * (NB * 16);

  // ---- B fragments: B[k][frame], k = 2d -> x_d^2, 2d+1 -> x_d, 2*dim -> 1, beyond -> 0 ----------
  // The tile's frames are one contiguous span of the feature buffer: copy it coalesced into LDS
  // (the stage buffers are free until the main loop starts), then let every lane pick its values.
  constexpr int kTileFrames = kWaves * NB * 16;
  static_assert((size_t)kTileFrames * (2 * KSTEPS) * sizeof(float) <= sizeof(lds), "feature tile must fit the stage buffers");
  double bf[NB][KSTEPS];
  {
    float* fl = reinterpret_cast<float*>(lds);
    const uint64_t tile_first = (uint64_t)x * kTileFrames;
    const uint64_t tile_frames = (a.n_frames - tile_first < (uint64_t)kTileFrames) ? a.n_frames - tile_first : kTileFrames;
    const uint32_t n_floats = (uint32_t)tile_frames * a.dim;
    const float* src = a.feats + tile_first * a.dim;
    // 16-byte loads from the 16-byte aligned part of the span (a chunk may start at any frame), dword
    // stores into LDS so that fl[i] == src[i] whatever the alignment; all loads issued before the first wait
    const uint32_t head = min((uint32_t)((4u - (uint32_t)((reinterpret_cast<uintptr_t>(src) >> 2) & 3u)) & 3u), n_floats);
    const uint32_t n4 = (n_floats - head) >> 2;
    if (n4 > 0) {
      constexpr int kMaxIter = (kTileFrames * 2 * KSTEPS / 4 + kWaves * 64 - 1) / (kWaves * 64);
      float4 tmp[kMaxIter];
#pragma unroll
      for (int j = 0; j < kMaxIter; j++) {
        const uint32_t i = threadIdx.x + j * (kWaves * 64);
        tmp[j] = reinterpret_cast<const float4*>(src + head)[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int j = 0; j < kMaxIter; j++) {
        const uint32_t i = threadIdx.x + j * (kWaves * 64);
        if (i < n4) {
          float* d = fl + head + 4 * i;
          d[0] = tmp[j].x; d[1] = tmp[j].y; d[2] = tmp[j].z; d[3] = tmp[j].w;
        }
      }
    }
    if (threadIdx.x < head) fl[threadIdx.x] = src[threadIdx.x];
    for (uint32_t i = head + (n4 << 2) + threadIdx.x; i < n_floats; i += kWaves * 64) fl[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const uint32_t fl_row = (uint32_t)wave * (NB * 16) + nb * 16 + c;
      const bool valid = fl_row < tile_frames;
      const float* xr = fl + (valid ? fl_row : 0u) * a.dim;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const uint32_t k = 4u * ks + g;
        const uint32_t d = k >> 1;
        const float xv = xr[d < a.dim ? d : 0u];
        const double xd = (valid && d < a.dim) ? (double)xv : 0.0;
        double v = (k & 1u) ? xd : xd * xd;
        if (valid && k == 2u * a.dim) v = 1.0;
        bf[nb][ks] = v;
      }
    }
    __syncthreads();
  }

  const double kInit = SUM ? 0.0 : 1e10;  // min_score seed (Mixtures.cpp:699) / sum_score seed (:721)
  double rm[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) rm[nb] = kInit;

  // ---- LDS staging by LDS-DMA: stage s holds blocks [b0 + s*kStageBlocks, ...) -------------------
  auto issue_stage = [&](uint32_t blk_first, int buf) {
#pragma unroll
    for (int i = 0; i < (kChunksPerStage + kWaves - 1) / kWaves; i++) {
      const int chunk = i * kWaves + wave;   // wave-uniform
      if (chunk < kChunksPerStage) {
        const uint32_t blk = blk_first + chunk / kChunksPerBlock;
        if (blk < b1) {
          const double* src = a.apack + (uint64_t)blk * kBlockDoubles + (chunk % kChunksPerBlock) * 128 + lane * 2;
          double* dst = lds + buf * kStageDoubles + chunk * 128;   // wave-uniform; HW adds lane*16
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    }
  };

  const uint32_t n_stages = (b1 - b0 + kStageBlocks - 1) / kStageBlocks;
  if (n_stages > 0) issue_stage(b0, 0);

  for (uint32_t s = 0; s < n_stages; s++) {
    const int buf = s & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70 | 0x0000);  // vmcnt(0): this wave's LDS-DMA pieces have landed
    __syncthreads();                              // everyone's pieces landed; buffer buf^1 is free again
    if (s + 1 < n_stages) issue_stage(b0 + (s + 1) * kStageBlocks, buf ^ 1);

#pragma unroll
    for (int bi = 0; bi < kStageBlocks; bi++) {
      const uint32_t blk = b0 + s * kStageBlocks + bi;
      if (blk < b1) {
        const double* ab = lds + buf * kStageDoubles + bi * kBlockDoubles + lane;
        v4d acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[nb] = (v4d){0.0, 0.0, 0.0, 0.0};
        double a_cur = ab[0];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          const double a_use = a_cur;
          if (ks + 1 < KSTEPS) a_cur = ab[(ks + 1) * 64];
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
            acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_use, bf[nb][ks], acc[nb], 0, 0, 0);
        }
        // mixture epilogue: four densities of state-slot g per lane
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double v = acc[nb][i];
            if (SUM) rm[nb] += exp(-v);
            else rm[nb] = (v < rm[nb]) ? v : rm[nb];   // strict <, NaN never wins (Mixtures.cpp:706)
          }
        }
        const uint32_t meta = a.blk_meta[__builtin_amdgcn_readfirstlane(blk)];
        if (meta & 1u) {  // last block of its 4-state group: emit and reset
          const uint32_t st = a.grp_state[4u * (meta >> 1) + g];
#pragma unroll
          for (int nb = 0; nb < NB; nb++) {
            const uint64_t f = frame0 + (uint64_t)nb * 16 + c;
            if (st != 0xFFFFFFFFu && f < a.n_frames) a.out[f * a.ld + st] = SUM ? -log(rm[nb]) : rm[nb];
            rm[nb] = kInit;
          }
        }
      }
    }
  }
}

template <int KSTEPS, int NB>
static hipError_t launch_k(const GmmMfmaArgs& a, bool sum, hipStream_t stream) {
  const dim3 grid(a.nx * a.ny), block(kWaves * 64);
  if (sum) hipLaunchKernelGGL((gmm_mfma_kernel<KSTEPS, NB, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((gmm_mfma_kernel<KSTEPS, NB, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

static int nb_for_ksteps(int ksteps) { return ksteps <= 20 ? 4 : 2; }
int gmm_mfma_frames_per_tile(int ksteps) { return kWaves * nb_for_ksteps(ksteps) * 16; }

int gmm_mfma_ksteps_for_dim(uint32_t dim) {
  const uint32_t k = 2 * dim + 1;
  if (k <= 32) return 8;
  if (k <= 56) return 14;
  if (k <= 80) return 20;
  if (k <= 128) return 32;
  return 0;
}

hipError_t launch_gmm_mfma(const GmmMfmaArgs& a, int ksteps, bool sum, hipStream_t stream) {
  switch (ksteps) {
    case 8: return launch_k<8, 4>(a, sum, stream);
    case 14: return launch_k<14, 4>(a, sum, stream);
    case 20: return launch_k<20, 4>(a, sum, stream);
    case 32: return launch_k<32, 2>(a, sum, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace srgpu
