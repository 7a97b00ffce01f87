// gmm_prefilter.hip -- exact max-approx GMM scoring with ~30x fewer FP64 flops: a bf16 MFMA prefilter that can
// only OVER-select, followed by an FP64 refinement of the selected densities in the reference's operation order.
//
// MixtureModel::min_score (sietill/Mixtures.cpp:696-713) needs, per (frame, state), only the minimum density
// score.  Kernel P computes every density score of the GEMM form  A[c,:].B[:,t]  (see gmm_mfma.hip) on the
// bf16 matrix cores with both operands split in two bf16 terms (x ~ x_hi + x_lo, products hi*hi + hi*lo + lo*hi,
// fp32 accumulation).  Its error is bounded:
//     |approx - exact| <= (3.03 * 2^-16 + 288 * 2^-24) * sum_k |a_k b_k| <= 6.4e-5 * |a|_2 |b|_2   =: eps / kKappaMargin
// (bf16 round-to-nearest keeps 8 significant bits per term, a 2-term split 16; the dropped lo*lo product and the
// two residuals are each <= 2^-16 |ab|; fp32 accumulation of <= 288 terms adds n * 2^-24).  With eps = 1e-4 |a||b|
// (1.5x margin; |a| = the largest coefficient norm of the state's densities, |b| computed per frame) every
// density whose approximation lies within 2*eps of the state's smallest approximation -- plus anything that is
// not a number -- is a candidate; the true arg-min is provably among them.  P writes one 32-bit candidate mask per
// (frame, state): 1.03-1.06 bits set on average.
//
// Kernel R evaluates only the candidates, in FP64, replaying density_score_sse's operation order
// (Mixtures.cpp:645-690): the minimum over the candidates is the minimum over all densities, bit for bit what
// MixtureModel::score returns.  P is MFMA/LDS-DMA bound (bf16), R is FP64-VALU/LDS bound with one density per
// (frame, state) instead of thirty-two.
//
// Limits of this path: max-approx only, <= 32 densities per mixture, dim <= 47 (K = 2*dim+1 <= 96); otherwise
// the API falls back to the FP64 MFMA kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace srgpu {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int kPWaves = 4;          // waves per workgroup (P)
static constexpr int kPStageBlocks = 4;    // 16-row model blocks per LDS stage
static constexpr int kGroupBlocks = 8;     // every 4-state group is padded to 8 blocks = 32 density slots per state
static constexpr float kKappa = 1.0e-4f;   // eps = kKappa * |a| * |b|   (bound derived above: 6.4e-5)

__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const __bf16 l = (__bf16)lo, h = (__bf16)hi;  // v_cvt_pk_bf16_f32: round to nearest even
  return (uint32_t)__builtin_bit_cast(uint16_t, l) | ((uint32_t)__builtin_bit_cast(uint16_t, h) << 16);
}
__device__ inline float bf16_round(float v) { return (float)(__bf16)v; }

// ---- kernel P ----------------------------------------------------------------------------------------------
// apack layout: [block][ks][part: 0 = hi, 1 = lo][lane][8 bf16]; row r of a block = density (r & 3) + 4*block_in_group of
// state slot (r >> 2), so that a lane's four accumulator registers (rows 4*(lane>>4) + reg) are four densities of ONE
// state slot for ONE frame (column lane & 15).
template <int KS32, int NB>
__global__ __launch_bounds__(kPWaves * 64) void gmm_prefilter_kernel(GmmPrefilterArgs a) {
  constexpr int kBlockBytes = KS32 * 2 * 1024;
  constexpr int kStageBytes = kPStageBlocks * kBlockBytes;
  constexpr int kChunksPerStage = kStageBytes / 1024;
  constexpr int kTileFrames = kPWaves * NB * 16;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kStageBytes];
  static_assert((size_t)kTileFrames * (16 * KS32) * sizeof(float) <= sizeof(lds), "feature tile (dim < 16*KS32) must fit the stage buffers");

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  uint32_t x, y;
  {
    const uint32_t id = blockIdx.x;
    if ((a.ny & 7u) == 0) { const uint32_t j = id >> 3; y = (id & 7u) + 8u * (j / a.nx); x = j % a.nx; }
    else { y = id / a.nx; x = id % a.nx; }
  }
  const uint32_t g0 = a.split_begin[y], g1 = a.split_begin[y + 1];  // group range of this workgroup
  const uint64_t frame0 = (uint64_t)x * kTileFrames + (uint64_t)wave * (NB * 16);

  // ---- B fragments (hi / lo bf16) and |b| per frame ---------------------------------------------------------
  uint4 bh[NB][KS32], bl[NB][KS32];
  float bnorm[NB];
  {
    float* fl = reinterpret_cast<float*>(lds);
    const uint64_t tile_first = (uint64_t)x * kTileFrames;
    const uint64_t tile_frames = (a.n_frames - tile_first < (uint64_t)kTileFrames) ? a.n_frames - tile_first : kTileFrames;
    const uint32_t n_floats = (uint32_t)tile_frames * a.dim;
    const float* src = a.feats + tile_first * a.dim;
    for (uint32_t i = threadIdx.x; i < n_floats; i += kPWaves * 64) fl[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const uint32_t row = (uint32_t)wave * (NB * 16) + nb * 16 + c;
      const bool valid = row < tile_frames;
      const float* xr = fl + (valid ? row : 0u) * a.dim;
      float n2 = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS32; ks++) {
        float hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t k = 32u * ks + 8u * g + j, d = k >> 1;
          const double xd = (valid && d < a.dim) ? (double)xr[d < a.dim ? d : 0u] : 0.0;
          double b = (k & 1u) ? xd : xd * xd;  // k = 2d -> x^2 (exact in double), 2d+1 -> x
          if (valid && k == 2u * a.dim) b = 1.0;
          const float h = bf16_round((float)b);
          hi[j] = h;
          lo[j] = (float)(b - (double)h);
          n2 += (float)(b * b);
        }
        bh[nb][ks] = make_uint4(pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]), pack_bf16x2(hi[4], hi[5]), pack_bf16x2(hi[6], hi[7]));
        bl[nb][ks] = make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(lo[4], lo[5]), pack_bf16x2(lo[6], lo[7]));
      }
      n2 += __shfl_xor(n2, 16);
      n2 += __shfl_xor(n2, 32);
      bnorm[nb] = sqrtf(n2) * 1.0001f;  // rounded up: fp32 summation of <= 96 positive terms
    }
    __syncthreads();
  }

  auto issue_stage = [&](uint32_t stage_first_block, int buf) {
#pragma unroll
    for (int i = 0; i < (kChunksPerStage + kPWaves - 1) / kPWaves; i++) {
      const int chunk = i * kPWaves + wave;
      if (chunk < kChunksPerStage) {
        const unsigned char* src = a.apack + (uint64_t)stage_first_block * kBlockBytes + (uint64_t)chunk * 1024 + lane * 16;
        unsigned char* dst = lds + buf * kStageBytes + chunk * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // stage index s covers blocks [g0*8 + 4s, +4): two stages per group
  const uint32_t n_stages = (g1 - g0) * (kGroupBlocks / kPStageBlocks);
  if (n_stages > 0) issue_stage(g0 * kGroupBlocks, 0);
  uint32_t s = 0;
  for (uint32_t grp = g0; grp < g1; grp++) {
    v4f ap[NB][kGroupBlocks];
#pragma unroll
    for (int half = 0; half < kGroupBlocks / kPStageBlocks; half++, s++) {
      const int buf = s & 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's LDS-DMA pieces have landed
      __syncthreads();
      if (s + 1 < n_stages) issue_stage(g0 * kGroupBlocks + (s + 1) * kPStageBlocks, buf ^ 1);
#pragma unroll
      for (int j = 0; j < kPStageBlocks; j++) {
        const uint4* blk = reinterpret_cast<const uint4*>(lds + buf * kStageBytes + j * kBlockBytes) + lane;
        v4f acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS32; ks++) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, blk[(ks * 2 + 0) * 64]);
          const bf16x8 al = __builtin_bit_cast(bf16x8, blk[(ks * 2 + 1) * 64]);
#pragma unroll
          for (int nb = 0; nb < NB; nb++) {
            const bf16x8 bhv = __builtin_bit_cast(bf16x8, bh[nb][ks]), blv = __builtin_bit_cast(bf16x8, bl[nb][ks]);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhv, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blv, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhv, acc[nb], 0, 0, 0);
          }
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) ap[nb][half * kPStageBlocks + j] = acc[nb];
      }
    }
    // ---- candidate mask of state slot g of this group, for the lane's frame(s) -------------------------------------
    const float na = a.grp_anorm[4u * grp + g];  // largest |a| over the state's densities (rounded up on the host)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      float amin = __builtin_huge_valf();
#pragma unroll
      for (int j = 0; j < kGroupBlocks; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) amin = fminf(amin, ap[nb][j][i]);  // fminf drops NaNs
      const float limit = amin + 2.0f * (kKappa * 1.001f) * na * bnorm[nb];
      uint32_t mask = 0;
#pragma unroll
      for (int j = 0; j < kGroupBlocks; j++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          mask |= (!(ap[nb][j][i] > limit) ? 1u : 0u) << (4 * j + i);  // NaN (bad variance) or inf limit: stay candidates
      const uint64_t f = frame0 + (uint64_t)nb * 16 + c;
      if (f < a.n_frames) a.mask[((uint64_t)grp * a.n_frames + f) * 4u + g] = mask;  // 256 contiguous bytes per wave
    }
  }
}

hipError_t launch_gmm_prefilter(const GmmPrefilterArgs& a, int ks32, hipStream_t stream) {
  const dim3 grid(a.nx * a.ny), block(kPWaves * 64);
  switch (ks32) {
    case 1: hipLaunchKernelGGL((gmm_prefilter_kernel<1, 2>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((gmm_prefilter_kernel<2, 2>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((gmm_prefilter_kernel<3, 2>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
int gmm_prefilter_frames_per_tile() { return kPWaves * 2 * 16; }

// ---- kernel R ----------------------------------------------------------------------------------------------
// One thread per frame (its feature vector converted to FP64 once and kept in registers), a workgroup walks the
// states of its range.  The FP64 rows of the current state -- [mu_0, 1/var_0, mu_1, 1/var_1, ..., norm, logw], row
// stride padded to an odd number of 16-byte pieces so that a wave's row gather spreads over all LDS banks -- are
// brought in by LDS-DMA one state ahead; every lane then reads the rows of ITS candidates.
#pragma clang fp contract(off)

static constexpr int kRThreads = 256;
static constexpr int kRWaves = kRThreads / 64;

template <int DT>  // DT = compile-time feature dimension (0: run-time a.dim, features re-read from featsT)
__global__ __launch_bounds__(kRThreads) void gmm_refine_kernel(GmmRefineArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char panel_raw[];  // [2][panel_bytes]
  const uint32_t D = DT ? (uint32_t)DT : a.dim, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t f = (uint64_t)blockIdx.x * kRThreads + tid;
  const bool valid = f < a.n_frames;
  const uint64_t fc = valid ? f : 0;
  const uint32_t s0 = blockIdx.y * a.states_per_split;
  const uint32_t s1 = (s0 + a.states_per_split < a.n_states) ? s0 + a.states_per_split : a.n_states;
  const uint32_t row_bytes = a.row_stride * 8u;
  const uint32_t panel_bytes = (a.max_dens * row_bytes + 1023u) & ~1023u;
  const uint32_t D2 = D - (D & 1u);

  double x[DT ? DT : 1];
  if (DT) {
#pragma unroll
    for (int k = 0; k < DT; k++) x[k] = (double)a.featsT[(uint64_t)k * a.n_frames_ld + fc];
  }
  auto X = [&](uint32_t k) -> double { return DT ? x[DT ? k : 0] : (double)a.featsT[(uint64_t)k * a.n_frames_ld + fc]; };

  // LDS-DMA of one state's rows: 1 KB per wave instruction, round-robin over the waves.  The last piece may run past
  // the state's rows (into the next state's, or into the tail padding of the buffer): never read back.
  auto issue_panel = [&](uint32_t st, int buf) {
    const uint32_t c0 = a.dens_off[st], n = a.dens_off[st + 1] - c0;
    const uint32_t chunks = __builtin_amdgcn_readfirstlane((n * row_bytes + 1023u) >> 10);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.rows) + (uint64_t)c0 * row_bytes + lane * 16;
    for (uint32_t ch = wave; ch < chunks; ch += kRWaves) {
      unsigned char* dst = panel_raw + (size_t)buf * panel_bytes + ch * 1024u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint64_t)ch * 1024u),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  uint32_t n_eval = 0;
  if (s0 < s1) issue_panel(s0, 0);
  // eight states per pass: a thread then owns one 64-byte aligned piece of its output row (ld and the split size
  // are multiples of 8) instead of scattering 8-byte stores that each cost a 64-byte HBM write
  for (uint32_t s8 = s0; s8 < s1; s8 += 8) {
    double res[8];
    uint32_t mk[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint4 v = (s8 + 4 * h < s1) ? reinterpret_cast<const uint4*>(a.mask)[(uint64_t)((s8 >> 2) + h) * a.n_frames + fc]
                                        : make_uint4(0, 0, 0, 0);
      mk[4 * h] = v.x; mk[4 * h + 1] = v.y; mk[4 * h + 2] = v.z; mk[4 * h + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t st = s8 + j;
      res[j] = 0.0;
      if (st < s1) {  // workgroup-uniform
        const int buf = (st - s0) & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of panel `buf` have landed
        __syncthreads();                     // ... and everybody's; panel buf^1 is no longer read
        if (st + 1 < s1) issue_panel(st + 1, buf ^ 1);
        const unsigned char* p = panel_raw + (size_t)buf * panel_bytes;
        const uint32_t n = a.dens_off[st + 1] - a.dens_off[st];
        uint32_t mask = mk[j];
        mask &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);  // padding slots of the prefilter are not densities
        n_eval += __builtin_popcount(mask);
        double best = 1e10;  // min_score seed (Mixtures.cpp:699)
        while (mask) {       // ascending density order, strict <: same minimum as the reference's full scan
          const uint32_t d = __builtin_ctz(mask);
          mask &= mask - 1;
          const double2* row = reinterpret_cast<const double2*>(p + d * row_bytes);  // row[k] = (mu_k, 1/var_k)
          double l0 = 0.0, l1 = 0.0;
#pragma unroll
          for (uint32_t k = 0; k < D2; k += 2) {
            const double2 r0 = row[k], r1 = row[k + 1];
            double u = X(k) - r0.x;
            u = u * u;
            u = u * r0.y;
            l0 = l0 + u;
            double v = X(k + 1) - r1.x;
            v = v * v;
            v = v * r1.y;
            l1 = l1 + v;
          }
          double dist = l0 + l1;
          if (D & 1u) {
            const double2 r = row[D - 1];
            const double t = X(D - 1) - r.x;
            dist += t * t * r.y;
          }
          const double2 nl = row[D];  // (norm, logw)
          double score = nl.x + dist / 2;
          score -= nl.y;
          if (score < best) best = score;
        }
        res[j] = best;
      }
    }
    if (valid) {
      double* o = a.out + f * a.ld + s8;
      if (s8 + 8 <= s1) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) *reinterpret_cast<double2*>(o + j) = make_double2(res[j], res[j + 1]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (s8 + j < s1) o[j] = res[j];
      }
    }
  }
  if (a.n_refined) {
    if (!valid) n_eval = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o);
    if (lane == 0) atomicAdd(a.n_refined, (unsigned long long)n_eval);
  }
}

__global__ void transpose_feats_kernel(const float* feats, uint64_t n_frames, uint32_t dim, uint64_t ldT, float* out) {
  __shared__ float tile[64][65];
  const uint64_t f0 = (uint64_t)blockIdx.x * 64;
  for (uint32_t d0 = 0; d0 < dim; d0 += 64) {
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
      const uint32_t fr = i / 64, d = i % 64;
      tile[fr][d] = (f0 + fr < n_frames && d0 + d < dim) ? feats[(f0 + fr) * dim + d0 + d] : 0.0f;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
      const uint32_t d = i / 64, fr = i % 64;
      if (f0 + fr < n_frames && d0 + d < dim) out[(uint64_t)(d0 + d) * ldT + f0 + fr] = tile[fr][d];
    }
    __syncthreads();
  }
}

hipError_t launch_transpose_feats(const float* feats, uint64_t n_frames, uint32_t dim, uint64_t ldT, float* out, hipStream_t stream) {
  if (n_frames == 0) return hipSuccess;
  hipLaunchKernelGGL(transpose_feats_kernel, dim3((unsigned)((n_frames + 63) / 64)), dim3(256), 0, stream, feats, n_frames, dim, ldT, out);
  return hipGetLastError();
}

hipError_t launch_gmm_refine(const GmmRefineArgs& a, uint32_t n_splits, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  const size_t smem = 2 * (((size_t)a.max_dens * a.row_stride * 8 + 1023) & ~(size_t)1023);
  const dim3 grid((unsigned)((a.n_frames + kRThreads - 1) / kRThreads), n_splits), block(kRThreads);
  auto go = [&](auto kernel) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, a);
    return hipGetLastError();
  };
  switch (a.dim) {
    case 39: return go(gmm_refine_kernel<39>);
    case 25: return go(gmm_refine_kernel<25>);
    default: return go(gmm_refine_kernel<0>);
  }
}

}  // namespace srgpu
