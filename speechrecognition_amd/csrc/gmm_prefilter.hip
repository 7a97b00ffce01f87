// gmm_prefilter.hip -- exact max-approx GMM scoring with ~30x fewer FP64 flops: an fp16 MFMA prefilter that can
// only OVER-select, followed by an FP64 refinement of the selected densities in the reference's operation order.
//
// MixtureModel::min_score (sietill/Mixtures.cpp:696-713) needs, per (frame, state), only the minimum density
// score.  Kernel P computes every density score in the GEMM form of gmm_mfma.hip,
//     score(c, t) = konst_c + sum_k a_ck b_kt,   a_c = [1/(2 var); -mu/var],  b_t = [x^2; x],
// on the fp16 matrix cores: both operands rounded ONCE to fp16 (11 significant bits), one MFMA per k-step, fp32
// accumulation; konst_c enters through three spare k slots as an exact 3-term fp16 expansion (33 bits) times 1.  The
// model side is scaled by a power of two sA (host: largest coefficient or constant -> below 2^14) so that fp16's
// narrow exponent range is used well; scores, norms and the candidate test stay in scaled units.  Error bound:
//   |delta a_k| <= 2^-11 |a_k| + 2^-25 / sA (subnormal spacing), |delta b_k| <= 2^-11 |b_k| + 2^-25; products of two
//   fp16 values are exact in fp32; <= 81 + 3 fp32 additions, each off by 2^-24 of a partial sum that never exceeds
//   |konst| + sum |a b| (first order):
//   |approx - exact| <= (2^-10 (1 + 2^-12) + 87 * 2^-24) |a||b| + 87 * 2^-24 |konst| + 2^-25 (|b|_1 / sA + |a|_1) (1 + 2^-11)
// -> eps = kKappa16 |a||b| + kKonst16 |konst| + kAbs16 (|b| / sA + |a|),  kKappa16 = 1.05e-3 (needed 9.82e-4: the
//    operand roundings are round-to-nearest by construction -- v_cvt_f16_f32 here, float -> _Float16 on the host, whose
//    double rounding adds 2^-13 relative to the 2^-11 -- so only the 87 * 2^-24 part rests on the hardware),
//    kKonst16 = 1.1e-5 (needed 5.19e-6 = 87 * 2^-24: x2, so that an accumulator that TRUNCATED every addition, 2^-23
//    each, would still be inside), kAbs16 = 6.0e-7 (needed 2^-25 (1 + 2^-11) sqrt(K) = 2.92e-7 at the padded K = 96,
//    |v|_1 <= sqrt(K) |v|_2: x2);  |a|, |konst| = the largest over the state's densities, rounded up on the host;
//    |b| per frame, rounded up.
// The accumulation model itself (<= 2^-24 of the running magnitude per addition, whatever the order inside the
// instruction) is probed on the device the model is created on, next to the subnormal probe: probe_fp16_accumulation()
// runs adversarial 96-term dot products with known exact sums through the same three-instruction MFMA chain and
// requires |error| <= 87 * 2^-24 * sum |a_k b_k|; a device that fails is scored by the exact kernel instead.
// Every density whose approximation lies within 2*eps of the state's smallest approximation -- plus anything that is
// not a number -- is a candidate; the true arg-min is provably among them.  A feature beyond fp16's range
// (|x| > 255) turns its frame's scores into inf/NaN: every density stays a candidate.  P writes one 32-bit candidate
// mask per (frame, state): 1.09 bits set on average on the bench model.
// (Measured alternative, removed again: both operands split in two bf16 terms, three products -- a 7x tighter bound,
// 1.01 candidates, but twice the prefilter time for 20 % less refinement time.)
//
// Kernel R evaluates only the candidates, in FP64, replaying density_score_sse's operation order
// (Mixtures.cpp:645-690): the minimum over the candidates is the minimum over all densities, bit for bit what
// MixtureModel::score returns.
//
// K = 128 (round 5, dim 47..62: four k-steps): <= 127 + 4 additions -> the same formula with 132 in place of 87:
//    kKappa16 needs 9.85e-4 (1.05e-3 stays), kKonst16 = 1.6e-5 (2 x 132 * 2^-24 = 1.57e-5), kAbs16 = 7.0e-7 (2 x 2^-25 (1 + 2^-11) sqrt(128)
//    = 6.75e-7); the accumulation probe runs its 128-term chain against 132 * 2^-24 on such a model's device.
//
// Limits of this path: max-approx only, <= 128 densities per mixture (a mixture of more than 32 spans 2 or 4
// consecutive 32-slot pseudo-states; up to 256 = 8 of them while the dimension is <= 39), dim <= 62 (K = 2*dim + 3 <= 128);
// any other model is scored by the exact FP64 kernel (same bits).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "kernels.h"

#ifndef SR_P16_NB
#define SR_P16_NB 4   // 16-frame column blocks per wave in the fp16 prefilter (4: 256 frames per workgroup, 2 waves/SIMD)
#endif

namespace srgpu {

typedef float v4f __attribute__((ext_vector_type(4)));
static constexpr int kPWaves = 4;          // waves per workgroup (P)
static constexpr int kGroupBlocks = 8;     // every 4-state group is padded to 8 blocks = 32 density slots per state

// ---- kernel P ----------------------------------------------------------------------------------------------
// apack layout: [group][8 blocks][k-step of 32][lane][8 fp16] = the A operand of v_mfma_f32_16x16x32_f16 (lane l holds
// row l & 15, k = 32*ks + 8*(l >> 4) + j); row r of a block = density (r & 3) + 4*block of state slot (r >> 2), so that
// a lane's four accumulator registers (rows 4*(lane>>4) + reg) are four densities of ONE state slot for ONE frame
// (column lane & 15).  One LDS stage = one group = 8 blocks, LDS-DMA double buffered; the A fragment of step t+1 is
// read before the MFMAs of step t issue, and at the last step of a stage the wave joins the barrier for the NEXT
// stage first, so that the pipeline runs across stage boundaries and the LDS-DMA of stage s+2 goes into the buffer
// everybody has just finished reading.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static constexpr float kKappa16 = 1.05e-3f;
template <int KS32> struct PfBound { static constexpr float kKonst16 = 1.1e-5f, kAbs16 = 6.0e-7f; };  // K <= 96 (file header)
template <> struct PfBound<4> { static constexpr float kKonst16 = 1.6e-5f, kAbs16 = 7.0e-7f; };        // K = 128

__device__ inline uint32_t pack_f16x2(float lo, float hi) {
  const _Float16 l = (_Float16)lo, h = (_Float16)hi;
  return (uint32_t)__builtin_bit_cast(uint16_t, l) | ((uint32_t)__builtin_bit_cast(uint16_t, h) << 16);
}

template <int KS32, int NB>
__global__ __launch_bounds__(kPWaves * 64, (NB <= 2 ? 3 : 2)) void gmm_prefilter16_kernel(GmmPrefilterArgs a) {
  constexpr int kBlockBytes = KS32 * 1024;
  constexpr int kStageBytes = kGroupBlocks * kBlockBytes;
  constexpr int kChunksPerStage = kStageBytes / 1024;
  constexpr int kDmaPerWave = kChunksPerStage / kPWaves;
  static_assert(kChunksPerStage % kPWaves == 0, "every wave issues the same number of LDS-DMA pieces per stage");
  constexpr int kTileFrames = kPWaves * NB * 16;
  constexpr size_t kTileBytes = (size_t)kTileFrames * (16 * KS32) * sizeof(float);
  constexpr size_t kLdsBytes = 2 * kStageBytes > kTileBytes ? 2 * kStageBytes : kTileBytes;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kLdsBytes];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  uint32_t x, y;
  {
    const uint32_t id = blockIdx.x;
    if ((a.ny & 7u) == 0) { const uint32_t j = id >> 3; y = (id & 7u) + 8u * (j / a.nx); x = j % a.nx; }
    else { y = id / a.nx; x = id % a.nx; }
  }
  const uint32_t g0 = a.split_begin[y], g1 = a.split_begin[y + 1];
  const uint64_t frame0 = (uint64_t)x * kTileFrames + (uint64_t)wave * (NB * 16);

  // ---- B fragments (fp16) and |b| per frame ---------------------------------------------------------------------
  uint4 bf[NB][KS32];
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
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t k = 32u * ks + 8u * g + j, d = k >> 1;
          const float xf = (valid && d < a.dim) ? xr[d < a.dim ? d : 0u] : 0.0f;
          float b = (k & 1u) ? xf : xf * xf;  // k = 2d -> x^2, 2d+1 -> x   (fp32 rounding of x^2: 2^-24, inside the margin)
          n2 += b * b;
          if (valid && k >= 2u * a.dim && k < 2u * a.dim + 3u) b = 1.0f;  // the three konst slots; not part of |b|
          v[j] = b;
        }
        bf[nb][ks] = make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
      }
      n2 += __shfl_xor(n2, 16);
      n2 += __shfl_xor(n2, 32);
      bnorm[nb] = sqrtf(n2) * 1.0001f;  // rounded up
    }
    __syncthreads();
  }

  // groups are fetched in order: a running per-lane source pointer (one 64-bit add per piece instead of two)
  const unsigned char* dma_src = a.apack + (uint64_t)g0 * kStageBytes + (uint64_t)wave * 1024 + lane * 16;
  auto issue_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kDmaPerWave; i++) {
      const int chunk = i * kPWaves + wave;
      unsigned char* dst = lds + buf * kStageBytes + chunk * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src + i * (kPWaves * 1024)),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    dma_src += kStageBytes;
  };
  auto frag = [&](int buf, int j, int ks) -> f16x8 {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(lds + buf * kStageBytes + j * kBlockBytes + ks * 1024 + lane * 16));
  };

  // one stage per group; A fragments are read one k-step ahead, across the stage boundary too
  const uint32_t n_stages = g1 - g0;
  if (n_stages > 0) issue_stage(0);
  if (n_stages > 1) issue_stage(1);
  if (n_stages > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | kDmaPerWave);  // vmcnt(kDmaPerWave): stage 0 has landed
  else __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  f16x8 a_c, a_n;
  if (n_stages > 0) a_c = frag(0, 0, 0);
  // where the lane's mask words go, relative to the group's first entry (4 n_frames < 2^32: the launcher checks)
  uint32_t* m_ptr = a.mask + (uint64_t)g0 * a.n_frames * 4u;
  uint32_t m_off[NB];
  bool m_ok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
    const uint64_t f = frame0 + (uint64_t)nb * 16 + c;
    m_ok[nb] = f < a.n_frames;
    m_off[nb] = (uint32_t)f * 4u + (uint32_t)g;
  }
  for (uint32_t s = 0; s < n_stages; s++) {
    const uint32_t grp = g0 + s;
    const int buf = s & 1;
    v4f ap[NB][kGroupBlocks];
#pragma unroll
    for (int j = 0; j < kGroupBlocks; j++) {
      v4f acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; nb++) acc[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS32; ks++) {
        if (j == kGroupBlocks - 1 && ks == KS32 - 1) {
          if (s + 1 < n_stages) {  // workgroup-uniform
            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): stage s+1 landed, my reads of stage s returned
            __syncthreads();
            if (s + 2 < n_stages) issue_stage(buf);
            a_n = frag(buf ^ 1, 0, 0);
          }
        } else {
          a_n = frag(buf, (ks == KS32 - 1) ? j + 1 : j, (ks == KS32 - 1) ? 0 : ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_c, __builtin_bit_cast(f16x8, bf[nb][ks]), acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_c = a_n;
      }
#pragma unroll
      for (int nb = 0; nb < NB; nb++) ap[nb][j] = acc[nb];
    }
    // ---- candidate mask of state slot g of this group, for the lane's frame(s) (scaled units) ---------------------
    const float2 nk = reinterpret_cast<const float2*>(a.grp_anorm)[4u * grp + g];  // sA |a|, sA |konst| (rounded up)
    // 2 eps = lim1 |b| + lim0 (the bound of the file header, the frame's part factored out: 2 instructions per frame block)
    constexpr float kKonst16 = PfBound<KS32>::kKonst16, kAbs16 = PfBound<KS32>::kAbs16;
    const float lim1 = 2.0f * (kKappa16 * nk.x + kAbs16), lim0 = 2.0f * (kKonst16 * nk.y + kAbs16 * nk.x);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      // v_min3_f32 directly: the hardware minimum already drops (quiet) NaNs; fminf() would canonicalise every operand
      // first (one v_max_f32 each).  Should a NaN survive, the limit is NaN and every density stays a candidate.
      // (one asm statement per two blocks: the compiler pads every asm statement with an s_nop)
      float amin = __builtin_huge_valf();
#pragma unroll
      for (int j = 0; j < kGroupBlocks; j += 2)
        asm("v_min3_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %3, %4\n\tv_min3_f32 %0, %0, %5, %6\n\tv_min3_f32 %0, %0, %7, %8"
            : "+v"(amin)
            : "v"(ap[nb][j][0]), "v"(ap[nb][j][1]), "v"(ap[nb][j][2]), "v"(ap[nb][j][3]), "v"(ap[nb][j + 1][0]),
              "v"(ap[nb][j + 1][1]), "v"(ap[nb][j + 1][2]), "v"(ap[nb][j + 1][3]));
      if (a.chunks >= 2) { const float o = __shfl_xor(amin, 16); asm("v_min_f32 %0, %0, %1" : "+v"(amin) : "v"(o)); }  // the other chunk(s) of the state
      if (a.chunks >= 4) { const float o = __shfl_xor(amin, 32); asm("v_min_f32 %0, %0, %1" : "+v"(amin) : "v"(o)); }
      const float limit = amin + __builtin_fmaf(lim1, bnorm[nb], lim0);
      uint32_t mask = 0;
#pragma unroll
      for (int j = kGroupBlocks - 1; j >= 1; j -= 2)  // densities in descending order, two blocks per asm statement
        asm("v_cmp_ngt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %3, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %4, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %5, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %6, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %7, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %8, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_ngt_f32 vcc, %9, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
            : "+v"(mask)
            : "v"(limit), "v"(ap[nb][j][3]), "v"(ap[nb][j][2]), "v"(ap[nb][j][1]), "v"(ap[nb][j][0]), "v"(ap[nb][j - 1][3]),
              "v"(ap[nb][j - 1][2]), "v"(ap[nb][j - 1][1]), "v"(ap[nb][j - 1][0])
            : "vcc");
      if (m_ok[nb]) m_ptr[m_off[nb]] = mask;
    }
    m_ptr += a.n_frames * 4u;  // the next group's entries (wave-uniform pointer, 32-bit lane offsets)
  }
}

// The error bound above counts fp16 subnormals as representable (spacing 2^-24).  hipcc's default kernel mode keeps
// them (float_denorm_mode_16_64 = preserve) and MFMA A/B inputs honour that mode; this probe checks it on the device
// the model is created on: 2^-20 (subnormal in fp16) x 2^10 must come out as 2^-10, not 0.
__global__ void fp16_denormal_probe_kernel(float* out) {
  const int lane = threadIdx.x;
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < 16) { a[0] = (_Float16)9.5367431640625e-7f; b[0] = (_Float16)1024.0f; }  // k = 0 of every row / column
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  out[lane] = acc[0];
}

hipError_t probe_fp16_denormals(hipStream_t stream, bool* preserved) {
  float* d = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), 64 * sizeof(float));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fp16_denormal_probe_kernel, dim3(1), dim3(64), 0, stream, d);
  float h[64];
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return e;
  *preserved = true;
  for (int i = 0; i < 64; i++) *preserved = *preserved && h[i] == 9.765625e-4f;
  return hipSuccess;
}

// ---- accumulation probe ------------------------------------------------------------------------------------------
// One 16x16 tile through the kernel's own MFMA chain (KS k-steps of 32, accumulator from zero): A[16][K] and B[K][16], K = 32 KS,
// fp16 from global memory, D[16][16] fp32 back.  The host builds the operands (below) and knows every exact sum.
template <int KS>
__global__ void fp16_accumulation_probe_kernel(const _Float16* A, const _Float16* B, float* out) {
  constexpr int K = 32 * KS;
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 32 * ks + 8 * q + j;
      a[j] = A[r * K + k];    // A operand: row = lane & 15, k = 32 ks + 8 (lane >> 4) + j
      b[j] = B[k * 16 + r];   // B operand: column = lane & 15, same k
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[(4 * q + i) * 16 + r] = acc[i];  // C/D: column = lane & 15, row = 4 (lane >> 4) + reg
}

// Adversarial operand sets, all products exact in fp32 (11 x 11 significant bits) and all exact sums exact in double:
//   family 0  one product of 2^23 and K - 1 products of 0.75: every addition to the big term is a tie-or-worse rounding
//             (at K = 96 a truncating accumulator ends about 50 units low, the model allows 43)
//   family 1  cancellation: K / 2 - 1 pairs +-(2^10 + d_k) with different small d_k, residue ~ 2^-4 .. 2^2 against sum |p| ~ 1e5
//   family 2  K equal products (1 + 2^-10)^2: a short internal accumulator or a wrong k order shows at once
//   family 3  random signs and magnitudes over 12 binades
// ks32 = 3 (K = 96) or 4 (K = 128).  worst_ratio (optional) receives max |error| / (2^-24 sum |p|) -- the model allows 87 / 132.
hipError_t probe_fp16_accumulation(hipStream_t stream, int ks32, bool* ok, double* worst_ratio) {
  if (ks32 != 3 && ks32 != 4) return hipErrorInvalidValue;
  const int kCases = 8;  // 8 tiles of 16 x 16 dot products
  const int K = 32 * ks32;
  const double allowed = ks32 == 3 ? 87.0 : 132.0;
  std::vector<_Float16> hA((size_t)kCases * 16 * K), hB((size_t)kCases * K * 16);
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  for (int t = 0; t < kCases; t++) {
    _Float16* A = hA.data() + (size_t)t * 16 * K;
    _Float16* B = hB.data() + (size_t)t * K * 16;
    for (int k = 0; k < K; k++)
      for (int c = 0; c < 16; c++) B[k * 16 + c] = (_Float16)1.0f;
    for (int r = 0; r < 16; r++)
      for (int k = 0; k < K; k++) {
        float a = 0.f;
        const int fam = (t * 16 + r) & 3;
        if (fam == 0) a = k == (r * 5) % K ? 4096.0f : 0.75f;                      // times b = 2048 / 1 below
        else if (fam == 1) a = k >= K - 2 ? 0.f : ((k & 1) ? -1.f : 1.f) * (1024.0f + (float)(((k >> 1) * 7 + r) % 64) * ((k & 1) ? 0.5f : 0.53125f));
        else if (fam == 2) a = 1.0009765625f;
        else { const uint64_t u = next(); a = ((u & 1) ? -1.f : 1.f) * (1.0f + (float)((u >> 8) & 1023) / 1024.0f) * (float)(1u << ((u >> 20) % 12)); }
        A[r * K + k] = (_Float16)a;
      }
    // column-side factors: column c scales family 0's big term to 2^23 and gives the others a second 11-bit factor
    for (int k = 0; k < K; k++)
      for (int c = 0; c < 16; c++) {
        const uint64_t u = next();
        float b = 1.0f + (float)((u >> 5) % 1024) / 1024.0f;                          // [1, 2): full 11-bit significand
        if ((c & 3) == 0) b = 1.0f;
        if ((c & 3) == 1) b = 1.0009765625f;
        B[k * 16 + c] = (_Float16)b;
      }
    for (int c = 0; c < 16; c++)  // family 0 rows meet their 2048 on one k only (any column): overwrite that k's factor
      for (int r = 0; r < 16; r++)
        if (((t * 16 + r) & 3) == 0) B[((r * 5) % K) * 16 + c] = (_Float16)2048.0f;
  }
  _Float16 *dA = nullptr, *dB = nullptr;
  float* dO = nullptr;
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&dA), hA.size() * 2)) != hipSuccess) return e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&dB), hB.size() * 2)) != hipSuccess) { (void)hipFree(dA); return e; }
  if ((e = hipMalloc(reinterpret_cast<void**>(&dO), (size_t)kCases * 256 * 4)) != hipSuccess) { (void)hipFree(dA); (void)hipFree(dB); return e; }
  std::vector<float> got((size_t)kCases * 256);
  e = hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  for (int t = 0; t < kCases && e == hipSuccess; t++) {
    if (ks32 == 3) hipLaunchKernelGGL(fp16_accumulation_probe_kernel<3>, dim3(1), dim3(64), 0, stream, dA + (size_t)t * 16 * K, dB + (size_t)t * K * 16, dO + (size_t)t * 256);
    else hipLaunchKernelGGL(fp16_accumulation_probe_kernel<4>, dim3(1), dim3(64), 0, stream, dA + (size_t)t * 16 * K, dB + (size_t)t * K * 16, dO + (size_t)t * 256);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO);
  if (e != hipSuccess) return e;
  double worst = 0.0;
  bool fine = true;
  for (int t = 0; t < kCases; t++)
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 16; c++) {
        double exact = 0.0, mag = 0.0;
        for (int k = 0; k < K; k++) {
          const double p = (double)(float)hA[((size_t)t * 16 + r) * K + k] * (double)(float)hB[((size_t)t * K + k) * 16 + c];
          exact += p;  // exact: every product has <= 22 significant bits within 2^-10 .. 2^24
          mag += p < 0 ? -p : p;
        }
        const double err = std::fabs((double)got[(size_t)t * 256 + r * 16 + c] - exact);
        const double ratio = err / (mag * 5.9604644775390625e-8);
        if (!(ratio <= allowed)) fine = false;  // (NaN fails)
        if (ratio > worst || ratio != ratio) worst = ratio;
      }
  *ok = fine;
  if (worst_ratio) *worst_ratio = worst;
  return hipSuccess;
}

hipError_t launch_gmm_prefilter(const GmmPrefilterArgs& a, int ks32, hipStream_t stream) {
  const dim3 grid(a.nx * a.ny), block(kPWaves * 64);
  if (a.n_frames >= (1ull << 30)) return hipErrorInvalidValue;  // (32-bit mask offsets inside a group)
  switch (ks32) {
    case 1: hipLaunchKernelGGL((gmm_prefilter16_kernel<1, SR_P16_NB>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((gmm_prefilter16_kernel<2, SR_P16_NB>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((gmm_prefilter16_kernel<3, SR_P16_NB>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((gmm_prefilter16_kernel<4, SR_P16_NB>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
int gmm_prefilter_frames_per_tile() { return kPWaves * SR_P16_NB * 16; }

// ---- kernel R ----------------------------------------------------------------------------------------------
// State-stationary: a workgroup owns SPW (8, or 4 for wide features) consecutive states, whose FP64 parameters fill
// the CU's LDS once (160 KiB for 8 states x 32 densities x 39 dimensions), and streams frames through its threads:
// one frame per thread at a time, the feature vector converted to FP64 and kept in registers, one 64-byte piece of
// the output row per frame.  No barrier and no refill inside the frame loop.
// LDS image per state: planes [mu_0 | 1/var_0 | mu_1 | 1/var_1 | ... | norm | logw] of NS density slots each, density d in
// slot d.  A lane that evaluates density d reads plane[p][d] with ds_read_b64, whose bank pair is d mod 32: lanes on different
// densities of ONE state never conflict and lanes on the same density share one broadcast read -- so every wave instruction
// below evaluates candidates of a single state.  (A bit of the candidate mask beyond the state's densities is cleared before
// it is used; the guard bit of ctz(mask | 2^31) can only be reached by a lane whose result is discarded, and slot 31 of a
// narrower image is still inside the workgroup's LDS.)
//
// Work layout (round 2).  About 7 % of the (frame, state) pairs have more than one candidate.  Letting the lanes that
// hold such pairs loop (round 1) cost 0.44 extra wave-evaluations per pair-wave for 0.087 extra candidates, because a
// wave waits for its unluckiest lane.  Now:
//   main pass   every lane evaluates the FIRST candidate of each of its 8 states (no divergence) and writes the row
//               piece; a lane whose pair has further candidates appends its frame to a small wave-private list of that
//               state (v_cmp ballot + mbcnt, one 4-byte store).
//   level 1     whenever a state's list holds 64 frames the wave evaluates the SECOND candidate of those 64 pairs --
//               one state, 64 different frames, fully dense -- and lowers the table entry where it wins.  The frames
//               are recent, so their features, masks and row pieces still sit in L2 / Infinity Cache.  Pairs with a
//               third candidate (1 % of all) move on to the level-2 list of the state.
//   level 2     64 such pairs at a time; the lanes loop over what is left of their masks (the only divergent code, on
//               about 1 % of the pairs).
//   lists are flushed at the end of the frame range.  The result does not depend on the order of evaluation: it is
//   the minimum under strict '<' seeded with 1e10 (Mixtures.cpp:696-713), a NaN never wins, and a score is never -0
//   (l0 starts at +0, so dist and norm + dist/2 are never -0, and a - b is -0 only for a = -0): equal scores have equal
//   bits.
#pragma clang fp contract(off)

#ifndef SR_R_THREADS
#define SR_R_THREADS 768
#endif
#ifndef SR_R_BATCH
#define SR_R_BATCH 4
#endif
static constexpr int kRThreads = SR_R_THREADS;   // 3 waves per SIMD; the register budget of 168 holds x in FP64 (78) + pipeline
static constexpr int kRThreadsWide = 512;        // padded dimension 55 / 63: 2 waves per SIMD, a budget of 256 registers (x in FP64: up to 126)
static constexpr int kRWaves = kRThreads / 64;   // (the ring workspace is sized for the larger workgroup)
static constexpr int kRBatch = SR_R_BATCH;       // dimensions per software-pipeline stage of the candidate evaluation
#ifndef SR_R_BATCH_WIDE
#define SR_R_BATCH_WIDE 8
#endif
static constexpr int kRBatchWide = SR_R_BATCH_WIDE;  // the same at 512 threads (2 waves per SIMD: fewer waves to hide an LDS read behind; 8 against 4:
                                                     // 2-3 % at dim 50 / 62, profiles/r5_cliffs.txt)
static constexpr uint32_t kRingEntries = 128;    // per (wave, level, state): < 64 pending + <= 64 appended per iteration
static constexpr uint32_t kRingLists = 2 * 8 * kRingEntries;  // entries per wave: [level][state][128]
static constexpr uint32_t kRingWave = 4 * kRingLists;         // u32 per wave: 16-byte entries
struct __attribute__((aligned(16))) RingEntry { uint32_t lf, mask; double score; };  // frame, candidates left, best score so far: one 16-byte store / load
struct __attribute__((packed, aligned(4))) RowPiece { float v[4]; };  // 16 bytes of a feature row (rows are 4-byte aligned)

// CH = 32-slot chunks (pseudo-states) per state: 1, or 2 / 4 for mixtures of up to 64 / 128 densities.  A state's chunks are
// consecutive panels of ONE workgroup (SPW is a multiple of CH and s0 of SPW), and the prefilter's candidate limit is relative
// to the STATE's minimum, so for most pairs all but one of a state's masks are empty (93 % at 64 densities on the bench model):
// the main pass evaluates ONE candidate per state -- the first of the state's first non-empty mask (round 4; until then it
// evaluated one per pseudo-state and dropped the results of the empty ones: half of the FP64 work at 64 densities) -- and
// everything else goes to the lists of the pseudo-state it belongs to.
// DT = the PADDED feature dimension, always odd (round 5; gmm_refine_padded_dim): density_score_sse sums dimensions 0 .. D - D % 2 - 1
// in pairs and adds an odd D's last dimension to l0 + l1 afterwards (Mixtures.cpp:651-675).  A model of any dimension D <= DT runs the
// DT instantiation on planes and features laid out as
//     pair dimensions 0 .. D - D % 2 - 1 | zeros up to DT - 2 | tail: dimension D - 1 if D is odd, else zeros
// because a padded dimension computes (0 - 0)^2 * 0 = +0.0 and l + 0.0 == l bit for bit (l is never -0; NaN and inf stay what they
// are), and an even D's empty tail adds +0.0 to dist = l0 + l1.  Rounds 2-4 had a run-time-dimension instantiation (DT = 0) for every
// dimension but 25 and 39 that re-read each feature from memory per evaluation; it is gone.
// NT = threads per workgroup: 768 while the features fit a 168-register budget (DT <= 47: 161 registers there), 512 beyond (2 waves per
// SIMD hide the LDS reads worse: dim 40 in the 47 instantiation took 27.6 ms at 512 threads, 21.2 at 768; profiles/r5_cliffs.txt).
// VS = 2 (round 5, mixtures of 129 .. 256 densities): a state is TWO halves of four chunks each.  The fp16 pass works on 4-panel groups and
// takes its candidate limit within a group, so each half has a first candidate of its own: the main pass evaluates one per half and
// stores the smaller; everything else is the four-chunk machinery (lists per panel, atomic minimum on the state's entry).
// DEFER (round 5, mixtures of more than 32 densities): the pairs with candidates left over are not worked off in this kernel -- a batch
// there is a cold dependent chain (list entry -> frame -> feature row -> evaluation) between main-pass iterations that have no registers
// to prefetch with: 13 of configs[4]'s 45 ms for 0.11 evaluations per pair -- but appended to per-wave, per-panel SEGMENTS in global
// memory (capacity a.defer_cap entries each, counts in a.defer_cnt), which gmm_drain_kernel works off afterwards with every wave on
// batches.  A segment that is full (never at the capacity the launcher sizes: a quarter of the wave's pairs) makes the lane evaluate its
// leftovers on the spot, from the features it still holds, into the score it is about to store.
template <int DT, int NS, int SPW, int CH, int NT = kRThreads, int VS = 1, bool DEFER = false>
__global__ __launch_bounds__(NT) void gmm_refine_kernel(GmmRefineArgs a) {
  static_assert(DT > 0 && (DT & 1), "padded dimension: odd");
  static_assert(!DEFER || CH > 1, "deferred lists: states of several chunks");
  constexpr int kRThreads = NT, kRWaves = NT / 64;  // (shadow the file-scope defaults)
  static_assert(CH == 1 || CH == 2 || CH == 4, "chunks per state (half)");
  static_assert(VS == 1 || (VS == 2 && CH == 4), "two halves only of four-chunk states");
  static_assert(SPW % (CH * VS) == 0 && (CH == 1 || SPW % 4 == 0), "a state's chunks live in one workgroup");
  extern __shared__ __attribute__((aligned(1024))) unsigned char panel_raw[];  // [SPW][state_bytes], 1 KB granular
  const uint32_t D = (uint32_t)DT, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t s0 = blockIdx.x * SPW;
  const uint32_t ns = (s0 + SPW <= a.n_pstates) ? SPW : a.n_pstates - s0;  // (pseudo-)states of this workgroup
  const uint32_t state_bytes = (2u * D + 2u) * NS * 8u;
  const uint64_t f_begin = (uint64_t)blockIdx.y * a.frames_per_split;
  const uint64_t f_end = (f_begin + a.frames_per_split < a.n_frames) ? f_begin + a.frames_per_split : a.n_frames;

  // fill: 1 KB per wave instruction, round-robin over the waves (the last piece may run into the next state's planes
  // or the buffer's tail slack: never read back)
  {
    const uint32_t chunks = (ns * state_bytes + 1023u) >> 10;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.rows) + (uint64_t)s0 * state_bytes + lane * 16;
    for (uint32_t ch = wave; ch < chunks; ch += kRWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint64_t)ch * 1024u),
                                       (__attribute__((address_space(3))) void*)(panel_raw + ch * 1024u), 16, 0, 0);
  }
  uint32_t nd[SPW];  // densities per state (wave-uniform)
#pragma unroll
  for (int j = 0; j < SPW; j++) nd[j] = (uint32_t)j < ns ? a.n_dens_ps[s0 + j] : 0u;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  if (f_begin >= f_end) return;  // (after the barrier: whole workgroup)

  uint32_t n_eval = 0;  // evaluations of this WAVE (profiling; a scalar: counted by ballots, no per-lane arithmetic)
  double x[DT];
  // featsT through a buffer descriptor: address = base + 4 * f (VGPR) + 4 * k * ld (SGPR, stepped by scalar adds) -- one
  // VGPR offset for all dimensions instead of 39 loop-invariant 64-bit row pointers that spill out of the SGPR file and
  // come back through v_readlane.  (dim * ld * 4 < 2^32: the launcher checks.)
  const __amdgpu_buffer_rsrc_t featsT_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.featsT), 0, (int)0xFFFFFFFFu, 0x00020000);
  const uint32_t ld4 = (uint32_t)a.n_frames_ld * 4u;
  auto load_x = [&](uint64_t f) __attribute__((always_inline)) {
    const uint32_t voff = (uint32_t)f * 4u;
#pragma unroll
    for (int k = 0; k < DT; k++)
      x[k] = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(featsT_rsrc, voff, (uint32_t)k * ld4, 0));
  };
  // the same from the row-major buffer (rows of DT floats: the corpus' own when its dimension is DT, else the padded copy the
  // transpose kernel writes): for the batches below, whose 64 lanes hold 64 unrelated frames -- a frame's features are 2 cache
  // lines there against one line per dimension in featsT (measured: 11 ms of L2 traffic)
  auto load_x_row = [&](uint64_t f) __attribute__((always_inline)) {
    const float* xr = a.feats + f * (uint64_t)DT;
#pragma unroll
    for (int q = 0; q < DT / 4; q++) {
      const RowPiece t = reinterpret_cast<const RowPiece*>(xr)[q];
#pragma unroll
      for (int i = 0; i < 4; i++) x[4 * q + i] = (double)t.v[i];
    }
#pragma unroll
    for (int k = DT - DT % 4; k < DT; k++) x[k] = (double)xr[k];
  };
  // score of the density whose plane column starts at LDS address `col0`, in density_score_sse's operation order
  auto evaluate = [&](const unsigned char* col0) __attribute__((always_inline)) -> double {
    // volatile: keeps every read a ds_read_b64 (256 B/clk, 64 banks); merged into ds_read2_b64 they would run at
    // half rate on 32 banks, where densities d and d + 16 collide
    const volatile __attribute__((address_space(3))) double* col =
        (const volatile __attribute__((address_space(3))) double*)col0;  // col[plane * NS]
    double l0 = 0.0, l1 = 0.0, dist, score;
    {
      constexpr int RB = NT == kRThreadsWide ? kRBatchWide : kRBatch;
      // software pipeline over batches of RB dimensions: the reads of batch b+1 are issued before the
      // arithmetic of batch b (volatile reads are not moved by the compiler, hence the explicit fences)
      constexpr int NB_ = (DT + 1 + RB - 1) / RB;  // plane pairs 0..DT-1 = dimensions, pair DT = (norm, logw)
      double pm[2][RB], pv[2][RB];
#pragma unroll
      for (int i = 0; i < RB; i++)
        if (i <= DT) { pm[0][i] = col[(2 * i) * NS]; pv[0][i] = col[(2 * i + 1) * NS]; }
#pragma unroll
      for (int b = 0; b < NB_; b++) {
        const int cur = b & 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RB; i++) {
          const int k = (b + 1) * RB + i;
          if (k <= DT) { pm[cur ^ 1][i] = col[(2 * k) * NS]; pv[cur ^ 1][i] = col[(2 * k + 1) * NS]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RB; i++) {
          const int k = b * RB + i;
          if (k < (int)(DT - (DT & 1))) {
            double u = x[k] - pm[cur][i];
            u = u * u;
            u = u * pv[cur][i];
            if (k & 1) l1 = l1 + u; else l0 = l0 + u;
          } else if (k == DT - 1) {  // odd dimension count: scalar tail (Mixtures.cpp:680-683)
            dist = l0 + l1;
            const double t = x[k] - pm[cur][i];
            dist += t * t * pv[cur][i];
          } else if (k == DT) {
            if (!(DT & 1)) dist = l0 + l1;
            score = pm[cur][i] + dist / 2;
            score -= pv[cur][i];
          }
        }
        // pins this batch's arithmetic between the read groups (ordered against the volatile reads); without it
        // the arithmetic sinks below all 80 reads and their 160 destination registers spill
        asm volatile("" : "+v"(l0), "+v"(l1));
      }
    }
    return score;
  };

  // min_score's `if (score < min_score)` from the seed 1e10 (Mixtures.cpp:699-708) for the first candidate: v_min_f64 returns the
  // other operand for a NaN, i.e. the seed, exactly what the strict comparison leaves; one instruction instead of a compare and
  // two selects
  const double seed = 1e10;
  auto seeded_min = [&](double score) -> double {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(score), "s"(seed));
    return r;
  };
  // ---- wave-private candidate lists (global memory, a few KB per wave: L1/L2 resident) -----------------------------
  // an entry = (frame, candidates still to evaluate, best score so far): the batches need no look-up in the mask array or
  // the score table, they only store to the table where a later candidate wins
  RingEntry* ring = reinterpret_cast<RingEntry*>(a.ring + ((uint64_t)(blockIdx.y * gridDim.x + blockIdx.x) * kRWaves + wave) * kRingWave);
  const bool counting = a.n_refined != nullptr;  // (kernel argument: a scalar branch around the bookkeeping)
  uint64_t cnt1 = 0, cnt2 = 0;  // eight 8-bit counters each (wave-uniform): frames pending per state, level 1 / level 2
  // DEFER: this wave's segments [SPW][defer_cap] and how many entries each holds (wave-uniform)
  const uint64_t wave_global = (uint64_t)(blockIdx.y * gridDim.x + blockIdx.x) * kRWaves + wave;
  RingEntry* seg = DEFER ? reinterpret_cast<RingEntry*>(a.defer) + wave_global * SPW * (uint64_t)a.defer_cap : nullptr;
  uint32_t dcnt[DEFER ? SPW : 1];
  if constexpr (DEFER) {
#pragma unroll
    for (int j = 0; j < SPW; j++) dcnt[j] = 0;
  }
  const uint64_t f_wave = f_begin + (uint64_t)wave * 64;
  auto frame_of = [&](uint32_t lf) -> uint64_t { return f_wave + (uint64_t)(lf >> 6) * kRThreads + (lf & 63u); };
  // (CH == 1 reads the shift from the kernel argument although it is 0 there, and keeps the run-time branches of the round-3
  // kernel below: with the shift folded to a constant the compiler schedules the SAME instruction mix 1.0 ms slower on
  // configs[2] -- 16.0 against 15.0 ms, A/B on one box, gpurun_out/r4_ab_refine_cfg3.txt; profiles/r3_refine_closed.txt has
  // seen that before: "the present order is a local optimum somebody found")
  const uint32_t chunk_shift = CH == 1 ? (a.chunks == 1 ? 0u : a.chunks == 2 ? 1u : 2u) : CH == 2 ? 1u : VS == 2 ? 3u : 2u;

  const uint32_t n_it = (uint32_t)((f_end - f_begin + kRThreads - 1) / kRThreads);
  for (uint32_t it = 0; it < n_it; it++) {
    const uint64_t fr = f_wave + (uint64_t)it * kRThreads + lane;
    const bool valid = fr < f_end;
    const uint64_t f = valid ? fr : f_end - 1;  // lanes past the end shadow the last frame; they store and append nothing
    const uint32_t lf = it * 64u + (uint32_t)lane;
    load_x(f);
    // main pass: the first candidate of every state -- one evaluation per state in every lane, no divergence
    constexpr int RS = SPW / (CH * VS);  // whole states of this workgroup
    double res[RS];
    if constexpr (CH == 1) {
#pragma unroll
      for (int h = 0; h < SPW / 4; h++) {
        const uint4 v = reinterpret_cast<const uint4*>(a.mask)[(uint64_t)((s0 >> 2) + h) * a.n_frames + f];
        const uint32_t mk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const int j = 4 * h + jj;
          const uint32_t n = nd[j];
          uint32_t mask = mk[jj];
          mask &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);  // padding slots of the prefilter are not densities
          res[j] = 1e10;  // min_score seed (Mixtures.cpp:699)
          if (n) {        // wave-uniform (a state without densities has an empty mask and keeps the seed)
            // (mask != 0 here: the prefilter keeps the approximate arg-min a candidate -- or everything, when the frame or the
            // state is not a number -- and padding slots score +inf)
            const double score = evaluate(panel_raw + (size_t)j * state_bytes + (uint32_t)__builtin_ctz(mask | 0x80000000u) * 8u);
            res[j] = mask != 0 ? seeded_min(score) : res[j];
            if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(valid));  // (mask != 0, see above)
          }
          const bool more = valid && (mask & (mask - 1)) != 0;
          const uint64_t b = __ballot(more);
          if (b) {  // wave-uniform
            const uint32_t c1 = (uint32_t)(cnt1 >> (8 * j)) & 0xFFu;
            const uint32_t pos = c1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (more) {
              ring[(uint32_t)j * kRingEntries + pos] = RingEntry{lf, mask & (mask - 1), res[j]};
            }
            cnt1 += (uint64_t)__builtin_popcountll(b) << (8 * j);
          }
        }
      }
      // (the chunks == 2 / 4 branches are never taken in this instantiation -- the launcher sends such models to CH = 2 / 4 --
      // but they are part of the schedule, see chunk_shift above)
      if (valid) {
        if (a.chunks == 1) {
          double* o = a.out + f * a.ld + s0;
          if (ns == SPW) {  // ld is a multiple of 8: 64-byte (32-byte for SPW = 4) aligned pieces
#pragma unroll
            for (int j = 0; j < SPW; j += 2) *reinterpret_cast<double2*>(o + j) = make_double2(res[j], res[j + 1]);
          } else {
#pragma unroll
            for (int j = 0; j < SPW; j++)
              if ((uint32_t)j < ns) o[j] = res[j];
          }
        } else if (a.chunks == 2) {
          double* o = a.out + f * a.ld + s0 / 2;
#pragma unroll
          for (int j = 0; j < SPW; j += 2) {
            const double m = res[j + 1] < res[j] ? res[j + 1] : res[j];
            if ((uint32_t)j < ns) o[j / 2] = m;
          }
        } else {  // 4
          double* o = a.out + f * a.ld + s0 / 4;
#pragma unroll
          for (int j = 0; j < SPW; j += 4) {
            double m = res[j + 1] < res[j] ? res[j + 1] : res[j];
            m = res[j + 2] < m ? res[j + 2] : m;
            m = res[j + 3] < m ? res[j + 3] : m;
            if ((uint32_t)j < ns) o[j / 4] = m;
          }
        }
      }
    } else {
      // appends the pairs of pseudo-state j whose lanes still hold candidates (`rest`) to j's level-1 list
      auto append = [&](int j, bool more, uint32_t rest, double& best) __attribute__((always_inline)) {
        const uint64_t b = __ballot(more);
        if (b) {  // wave-uniform
          if constexpr (DEFER) {
            const uint32_t nb = (uint32_t)__builtin_popcountll(b);
            if (dcnt[DEFER ? j : 0] + nb <= a.defer_cap) {  // wave-uniform
              const uint32_t pos = dcnt[DEFER ? j : 0] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
              if (more) seg[(uint64_t)j * a.defer_cap + pos] = RingEntry{(uint32_t)f, rest, best};  // (the frame itself: n_frames < 2^30)
              dcnt[DEFER ? j : 0] += nb;
            } else {  // segment full: here and now, into the score the lane is about to store (the state's minimum so far)
              uint32_t mm = more ? rest : 0u;
              const unsigned char* panel = panel_raw + (size_t)j * state_bytes;
              do {
                const double score = evaluate(panel + (uint32_t)__builtin_ctz(mm | 0x80000000u) * 8u);
                if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(mm != 0));
                if (mm != 0 && score < best) best = score;
                mm &= mm - 1;
              } while (__any(mm != 0));
            }
          } else {
          const uint32_t c1 = (uint32_t)(cnt1 >> (8 * j)) & 0xFFu;
          const uint32_t pos = c1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
          if (more) {
            ring[(uint32_t)j * kRingEntries + pos] = RingEntry{lf, rest, best};
          }
          cnt1 += (uint64_t)__builtin_popcountll(b) << (8 * j);
          }
        }
      };
      uint32_t mk[SPW];
#pragma unroll
      for (int h = 0; h < SPW / 4; h++) {
        const uint4 v = reinterpret_cast<const uint4*>(a.mask)[(uint64_t)((s0 >> 2) + h) * a.n_frames + f];
        mk[4 * h] = v.x; mk[4 * h + 1] = v.y; mk[4 * h + 2] = v.z; mk[4 * h + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < SPW; j++) mk[j] &= nd[j] >= 32 ? 0xFFFFFFFFu : ((1u << nd[j]) - 1u);  // padding slots are not densities
      if constexpr (VS == 1) {  // (textually round 4's loop: the two-halves form below compiled 2 % slower at configs[4], profiles/r5_cliffs.txt)
#pragma unroll
      for (int r = 0; r < RS; r++) {
        // the chunk that holds the lane's first candidate: the first non-empty mask of the state (min_score's scan order,
        // Mixtures.cpp:700-708: densities ascending; the order does not matter for the result, see above)
        uint32_t msel = mk[r * CH], off = 0;
#pragma unroll
        for (int c = 1; c < CH; c++) {
          const bool empty = msel == 0;
          msel = empty ? mk[r * CH + c] : msel;
          off = empty ? (uint32_t)c * state_bytes : off;
        }
        res[r] = 1e10;  // min_score seed (Mixtures.cpp:699)
        if (nd[r * CH]) {  // wave-uniform (a state without densities keeps the seed; chunk 0 fills first)
          // lanes of one wave instruction now read up to CH panels, and slot d of every panel of a state shares a bank pair: 2-way
          // conflicts that cost nothing measurable (the LDS array is 41 % busy; profiles/r4_refine_chunked.txt, probe 1)
          const double score = evaluate(panel_raw + (size_t)(r * CH) * state_bytes + off + (uint32_t)__builtin_ctz(msel | 0x80000000u) * 8u);
          res[r] = msel != 0 ? seeded_min(score) : res[r];
          if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(valid));
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
          // what is left for the lists: the selected chunk without its first candidate; every later chunk in full (the
          // earlier ones are empty).  The entry carries the state's best score so far; the batches lower the table entry, which
          // the state's chunks share, with an atomic minimum where their candidate beats it.
          const bool is_sel = off == (uint32_t)c * state_bytes;
          const uint32_t rest = is_sel ? msel & (msel - 1) : mk[r * CH + c];
          if (nd[r * CH + c]) append(r * CH + c, valid && rest != 0, rest, res[r]);
        }
      }
      } else {
#pragma unroll
      for (int r = 0; r < RS; r++) res[r] = 1e10;  // min_score seed (Mixtures.cpp:699)
#pragma unroll
      for (int rv = 0; rv < RS * VS; rv++) {  // rv: a state, or one half of a state (VS = 2)
        // the chunk that holds the lane's first candidate: the first non-empty mask of the state (min_score's scan order,
        // Mixtures.cpp:700-708: densities ascending; the order does not matter for the result, see above)
        uint32_t msel = mk[rv * CH], off = 0;
#pragma unroll
        for (int c = 1; c < CH; c++) {
          const bool empty = msel == 0;
          msel = empty ? mk[rv * CH + c] : msel;
          off = empty ? (uint32_t)c * state_bytes : off;
        }
        double resv = 1e10;
        if (nd[rv * CH]) {  // wave-uniform (a state / half without densities keeps the seed; chunk 0 fills first)
          // lanes of one wave instruction now read up to CH panels, and slot d of every panel of a state shares a bank pair: 2-way
          // conflicts that cost nothing measurable (the LDS array is 41 % busy; profiles/r4_refine_chunked.txt, probe 1)
          const double score = evaluate(panel_raw + (size_t)(rv * CH) * state_bytes + off + (uint32_t)__builtin_ctz(msel | 0x80000000u) * 8u);
          resv = msel != 0 ? seeded_min(score) : resv;
          if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(valid));
        }
        if constexpr (VS == 1) res[rv] = resv;
        else res[rv / VS] = resv < res[rv / VS] ? resv : res[rv / VS];  // (equal scores have equal bits: file header)
#pragma unroll
        for (int c = 0; c < CH; c++) {
          // what is left for the lists: the selected chunk without its first candidate; every later chunk in full (the
          // earlier ones are empty).  The entry carries the best score so far of the state (of its half); the batches lower the table
          // entry, which the state's chunks share, with an atomic minimum where their candidate beats it.
          const bool is_sel = off == (uint32_t)c * state_bytes;
          const uint32_t rest = is_sel ? msel & (msel - 1) : mk[rv * CH + c];
          if (nd[rv * CH + c]) append(rv * CH + c, valid && rest != 0, rest, resv);
        }
        if constexpr (DEFER && VS == 2) res[rv / VS] = resv < res[rv / VS] ? resv : res[rv / VS];  // (a full segment lowered it in place)
      }
      }
      if (valid) {
        double* o = a.out + f * a.ld + s0 / (CH * VS);
        if (ns == SPW && RS >= 2) {  // ld is a multiple of 8 and s0 / CH one of RS: 16-byte aligned pairs
#pragma unroll
          for (int r = 0; r + 1 < RS; r += 2) *reinterpret_cast<double2*>(o + r) = make_double2(res[r], res[r + 1]);
        } else {
#pragma unroll
          for (int r = 0; r < RS; r++)
            if ((uint32_t)(r * CH * VS) < ns) o[r] = res[r];
        }
      }
    }
    // ---- work off the lists: every list is kept below 64 pending frames (one more iteration's appends must fit);
    // after the last iteration they are emptied.  One batch = <= 64 pairs of ONE state j (wave-uniform level, j, n):
    // level 1 evaluates each pair's second candidate, level 2 everything after the second.
    const bool last = it + 1 == n_it;
    if constexpr (!DEFER)
    for (;;) {
      uint32_t level, j, n;
      {
        const uint64_t full1 = cnt1 & 0x4040404040404040ull, full2 = cnt2 & 0x4040404040404040ull;
        // a full level-2 list first (level 1 appends to it), then a full level-1 list; at the end whatever is left
        const uint64_t pick2 = full2 ? full2 : (last && !full1 && !cnt1) ? cnt2 : 0;
        const uint64_t pick1 = full2 ? 0 : full1 ? full1 : last ? cnt1 : 0;
        const uint64_t pick = pick1 ? pick1 : pick2;
        if (!pick) break;
        level = pick1 ? 1u : 2u;
        j = (uint32_t)__builtin_ctzll(pick) >> 3;
        const uint32_t have_ = (uint32_t)((pick1 ? cnt1 : cnt2) >> (8u * j)) & 0xFFu;
        n = have_ < 64u ? have_ : 64u;
      }
      // opaque to the optimiser from here on: ONE copy of the batch code, whatever path selected it (threading the
      // selection through duplicated the evaluation 20 times and the kernel no longer fitted the instruction cache)
      asm volatile("" : "+s"(level), "+s"(j), "+s"(n));
      const bool l1 = level == 1;
      const uint32_t have = (uint32_t)((l1 ? cnt1 : cnt2) >> (8u * j)) & 0xFFu;
      const uint32_t slot0 = ((level - 1u) * 8u + j) * kRingEntries;
      if (l1) cnt1 -= (uint64_t)n << (8u * j); else cnt2 -= (uint64_t)n << (8u * j);
      const bool live = (uint32_t)lane < n;
      // The entries were stored by OTHER LANES OF THIS WAVE (the main pass above, a level-1 batch): a hand-off at wavefront scope.
      // Round 5: ordered by the language's own construct instead of a hand-counted `s_waitcnt vmcnt(N)` (rounds 3-4 waited for all
      // but the N youngest stores, on the claims that those are table stores and that stores complete in issue order -- true, and
      // invisible to the compiler: a recompile that split a store or sank a load would have broken it silently).  On
      // gfx950 a wave's vector-memory operations reach a given address in issue order, so the compiler emits NO wait for a
      // wavefront-scope release / acquire pair (tests/test_isa_cpu.py pins that lowering); should a target ever need one, the
      // fences are where it goes.  The asm barrier keeps the (non-atomic) entry accesses on their sides of the pair.
      asm volatile("" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      asm volatile("" ::: "memory");
      const uint32_t at = slot0 + have - n + (live ? (uint32_t)lane : 0u);
      const RingEntry en = ring[at];
      const uint32_t lfb = en.lf;
      uint32_t m = live ? en.mask : 0u;
      double cur = en.score;
      const uint64_t fb = frame_of(lfb);
      load_x_row(fb);
      double* o = a.out + fb * a.ld + ((s0 + j) >> chunk_shift);
      // A mixture of more than 32 densities: the table entry is shared by the state's pseudo-states and may have been lowered
      // through another one since the pair was listed.  Round 4: the batch does not read it back (round 3 did: a dependent
      // global round trip behind a full vmcnt(0) drain per batch -- 12.7 us per batch against 5 us for 32-density states, 17 of
      // configs[4]'s 49 ms, gpurun_out/r4_ab_refine_probe_cfg5.txt) -- it lowers the entry with a floating-point atomic minimum
      // where its candidate beats the score the pair was listed with: the entry ends as the minimum over all candidates whatever
      // the order (a NaN score fails `score < cur` and is never sent; scores are never -0).
      if constexpr (CH == 1) {
        if (chunk_shift) cur = *o;  // (never taken in this instantiation; part of the round-3 schedule, see chunk_shift)
      }
      const double before = cur;
      const unsigned char* panel = panel_raw + (size_t)j * state_bytes;
      do {  // level 1: exactly one round (every live lane has a second candidate); level 2: until every lane is done
        const double score = evaluate(panel + (uint32_t)__builtin_ctz(m | 0x80000000u) * 8u);
        if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(m != 0));
        if (m != 0 && score < cur) cur = score;
        m &= m - 1;
      } while (!l1 && __any(m != 0));
      if (l1) {
        const uint64_t b = __ballot(m != 0);
        if (b) {  // wave-uniform: pairs with a third candidate move on to level 2
          const uint32_t c2 = (uint32_t)(cnt2 >> (8u * j)) & 0xFFu;
          const uint32_t pos = c2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
          if (m != 0) {
            ring[(8u + j) * kRingEntries + pos] = RingEntry{lfb, m, cur};
          }
          cnt2 += (uint64_t)__builtin_popcountll(b) << (8u * j);
        }
      }
      // the table update comes last
      asm volatile("" ::: "memory");
      const bool lower = live && cur < before;
      if constexpr (CH == 1) {
        if (lower) *o = cur;
      } else {
        // global_atomic_min_f64 without a return value: issued behind this wave's earlier stores to the same entry (one wave owns a
        // frame's row pieces and lists), from another lane: the same wavefront-scope ordering as the list hand-off above
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lower) (void)__builtin_amdgcn_global_atomic_fmin_f64((__attribute__((address_space(1))) double*)o, cur);
      }
    }
  }
  if constexpr (DEFER) {
#pragma unroll
    for (int j = 0; j < SPW; j++)
      if (lane == 0) a.defer_cnt[wave_global * SPW + j] = dcnt[j];
  }
  if (a.n_refined) {
    if (lane == 0) atomicAdd(a.n_refined, (unsigned long long)n_eval);
  }
}

// ---- the deferred leftovers (DEFER above): every wave on batches -----------------------------------------------------------------------
// Same grid, same panels in LDS, same wave numbering as the gmm_refine_kernel<.., DEFER> launch before it: wave w of workgroup g works
// off the segments that wave wrote.  A batch = 64 entries of ONE panel: the lanes fetch their entries' feature rows (the next batch's
// entries are already in flight), evaluate each pair's next candidate and lower the table entry with an atomic minimum where it beats
// the score the pair was listed with; pairs with more candidates move on to the wave's small level-2 list of the panel (gmm_refine_kernel's
// ring), worked off 64 at a time, lanes looping over what is left of their masks.
template <int DT, int NS, int SPW, int CH, int NT, int VS>
__global__ __launch_bounds__(NT) void gmm_drain_kernel(GmmRefineArgs a) {
  constexpr int kRWaves = NT / 64;
  constexpr int RB = kRBatch;
  extern __shared__ __attribute__((aligned(1024))) unsigned char panel_raw[];
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t s0 = blockIdx.x * SPW;
  const uint32_t ns = (s0 + SPW <= a.n_pstates) ? SPW : a.n_pstates - s0;
  const uint32_t state_bytes = (2u * DT + 2u) * NS * 8u;
  {
    const uint32_t chunks = (ns * state_bytes + 1023u) >> 10;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.rows) + (uint64_t)s0 * state_bytes + lane * 16;
    for (uint32_t ch = wave; ch < chunks; ch += kRWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint64_t)ch * 1024u),
                                       (__attribute__((address_space(3))) void*)(panel_raw + ch * 1024u), 16, 0, 0);
  }
  const uint64_t wave_global = (uint64_t)(blockIdx.y * gridDim.x + blockIdx.x) * kRWaves + wave;
  const RingEntry* seg = reinterpret_cast<const RingEntry*>(a.defer) + wave_global * SPW * (uint64_t)a.defer_cap;
  RingEntry* ring = reinterpret_cast<RingEntry*>(a.ring + wave_global * kRingWave) + 8u * kRingEntries;  // the level-2 lists
  uint32_t nj[SPW];
#pragma unroll
  for (int j = 0; j < SPW; j++) nj[j] = (uint32_t)j < ns ? (uint32_t)__builtin_amdgcn_readfirstlane((int)a.defer_cnt[wave_global * SPW + j]) : 0u;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  const bool counting = a.n_refined != nullptr;
  uint32_t n_eval = 0;
  constexpr uint32_t chunk_shift = CH == 2 ? 1u : VS == 2 ? 3u : 2u;
  uint64_t cnt2 = 0;  // eight 8-bit counters: pairs pending per panel, level 2

  float xf[DT];
  auto load_row = [&](uint64_t f) __attribute__((always_inline)) {
    const float* xr = a.feats + f * (uint64_t)DT;
#pragma unroll
    for (int q = 0; q < DT / 4; q++) {
      const RowPiece t = reinterpret_cast<const RowPiece*>(xr)[q];
#pragma unroll
      for (int i = 0; i < 4; i++) xf[4 * q + i] = t.v[i];
    }
#pragma unroll
    for (int k = DT - DT % 4; k < DT; k++) xf[k] = xr[k];
  };
  // density_score_sse's operation order (Mixtures.cpp:645-690), as gmm_refine_kernel's evaluate(); the features are converted where
  // they are used (float registers: room for the next batch's entries and rows in flight)
  auto evaluate = [&](const unsigned char* col0) __attribute__((always_inline)) -> double {
    const volatile __attribute__((address_space(3))) double* col = (const volatile __attribute__((address_space(3))) double*)col0;
    double l0 = 0.0, l1 = 0.0, dist, score;
    constexpr int NB_ = (DT + 1 + RB - 1) / RB;
    double pm[2][RB], pv[2][RB];
#pragma unroll
    for (int i = 0; i < RB; i++)
      if (i <= DT) { pm[0][i] = col[(2 * i) * NS]; pv[0][i] = col[(2 * i + 1) * NS]; }
#pragma unroll
    for (int b = 0; b < NB_; b++) {
      const int cur = b & 1;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const int k = (b + 1) * RB + i;
        if (k <= DT) { pm[cur ^ 1][i] = col[(2 * k) * NS]; pv[cur ^ 1][i] = col[(2 * k + 1) * NS]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const int k = b * RB + i;
        if (k < DT - 1) {
          double u = (double)xf[k < DT ? k : 0] - pm[cur][i];
          u = u * u;
          u = u * pv[cur][i];
          if (k & 1) l1 = l1 + u; else l0 = l0 + u;
        } else if (k == DT - 1) {  // the odd tail (DT is odd)
          dist = l0 + l1;
          const double t = (double)xf[k < DT ? k : 0] - pm[cur][i];
          dist += t * t * pv[cur][i];
        } else if (k == DT) {
          score = pm[cur][i] + dist / 2;
          score -= pv[cur][i];
        }
      }
      asm volatile("" : "+v"(l0), "+v"(l1));
    }
    return score;
  };
  // one batch: the live lanes' entries `en` of panel j (wave-uniform); level 1: one candidate each, level 2: all that is left
  auto batch = [&](const bool l1, const uint32_t j, const RingEntry en, const bool live) __attribute__((always_inline)) {
    uint32_t m = live ? en.mask : 0u;
    double cur = en.score;
    const uint64_t fb = en.lf;
    load_row(fb);
    double* o = a.out + fb * a.ld + ((s0 + j) >> chunk_shift);
    const double before = cur;
    const unsigned char* panel = panel_raw + (size_t)j * state_bytes;
    do {
      const double score = evaluate(panel + (uint32_t)__builtin_ctz(m | 0x80000000u) * 8u);
      if (counting) n_eval += (uint32_t)__builtin_popcountll(__ballot(m != 0));
      if (m != 0 && score < cur) cur = score;
      m &= m - 1;
    } while (!l1 && __any(m != 0));
    if (l1) {
      const uint64_t b = __ballot(m != 0);
      if (b) {
        const uint32_t c2 = (uint32_t)(cnt2 >> (8u * j)) & 0xFFu;
        const uint32_t pos = c2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        if (m != 0) ring[j * kRingEntries + pos] = RingEntry{en.lf, m, cur};
        cnt2 += (uint64_t)__builtin_popcountll(b) << (8u * j);
      }
    }
    // (the entry was stored by the launch before this one; candidates of one pair reach it through atomics only: any order)
    if (live && cur < before) (void)__builtin_amdgcn_global_atomic_fmin_f64((__attribute__((address_space(1))) double*)o, cur);
  };
  auto level2 = [&](const bool flush) __attribute__((always_inline)) {  // full level-2 lists (at the end: every one)
    for (;;) {
      const uint64_t pick = (cnt2 & 0x4040404040404040ull) ? (cnt2 & 0x4040404040404040ull) : flush ? cnt2 : 0ull;
      if (!pick) break;
      uint32_t j = (uint32_t)__builtin_ctzll(pick) >> 3;
      asm volatile("" : "+s"(j));
      const uint32_t have = (uint32_t)(cnt2 >> (8u * j)) & 0xFFu, n = have < 64u ? have : 64u;
      cnt2 -= (uint64_t)n << (8u * j);
      const bool live = (uint32_t)lane < n;
      // entries stored by other lanes of this wave (level-1 batches above): wavefront-scope hand-off, as in gmm_refine_kernel
      asm volatile("" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      asm volatile("" ::: "memory");
      const RingEntry en = ring[j * kRingEntries + have - n + (live ? (uint32_t)lane : 0u)];
      batch(false, j, en, live);
    }
  };
#pragma unroll 1
  for (uint32_t j = 0; j < ns; j++) {
    uint32_t n_left = 0;
#pragma unroll
    for (int q = 0; q < SPW; q++) n_left = (uint32_t)q == j ? nj[q] : n_left;
    const RingEntry* sj = seg + (uint64_t)j * a.defer_cap;
    RingEntry nxt = n_left ? sj[(uint32_t)lane < n_left ? lane : 0] : RingEntry{0u, 0u, 0.0};
#pragma unroll 1
    for (uint32_t pos = 0; pos < n_left; pos += 64u) {
      const RingEntry en = nxt;
      const uint32_t n = n_left - pos < 64u ? n_left - pos : 64u;
      const uint32_t p2 = pos + 64u;
      if (p2 < n_left) nxt = sj[p2 + ((uint32_t)lane < n_left - p2 ? lane : 0)];  // in flight while this batch is evaluated
      batch(true, j, en, (uint32_t)lane < n);
      level2(false);
    }
  }
  level2(true);
  if (a.n_refined) {
    if (lane == 0) atomicAdd(a.n_refined, (unsigned long long)n_eval);
  }
}

// featsT[k][f] for the refinement's main pass, in the PADDED dimension order of gmm_refine_kernel (dp rows: the pair dimensions, zero
// rows, the odd tail last), and -- when dp differs from the corpus' dimension -- the same rows of dp floats row-major (featsP) for the
// batches; with dp == dim the batches read the corpus' own rows and featsP is null.
__device__ inline uint32_t padded_source_dim(uint32_t k, uint32_t dim, uint32_t dp) {  // which real dimension sits in padded slot k (dim: none)
  const uint32_t pairs = dim - (dim & 1u);
  return k < pairs ? k : (k == dp - 1u && (dim & 1u)) ? dim - 1u : dim;
}
__global__ void transpose_feats_kernel(const float* feats, uint64_t n_frames, uint32_t dim, uint32_t dp, uint64_t ldT, float* out, float* featsP) {
  __shared__ float tile[64][65];
  const uint64_t f0 = (uint64_t)blockIdx.x * 64;
  for (uint32_t d0 = 0; d0 < dp; d0 += 64) {
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
      const uint32_t fr = i / 64, d = i % 64;
      const uint32_t src = d0 + d < dp ? padded_source_dim(d0 + d, dim, dp) : dim;
      tile[fr][d] = (f0 + fr < n_frames && src < dim) ? feats[(f0 + fr) * dim + src] : 0.0f;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
      const uint32_t d = i / 64, fr = i % 64;
      if (f0 + fr < n_frames && d0 + d < dp) out[(uint64_t)(d0 + d) * ldT + f0 + fr] = tile[fr][d];
    }
    if (featsP) {
      for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const uint32_t fr = i / 64, d = i % 64;
        if (f0 + fr < n_frames && d0 + d < dp) featsP[(f0 + fr) * dp + d0 + d] = tile[fr][d];
      }
    }
    __syncthreads();
  }
}

hipError_t launch_transpose_feats(const float* feats, uint64_t n_frames, uint32_t dim, uint32_t dp, uint64_t ldT, float* out, float* featsP,
                                  hipStream_t stream) {
  if (n_frames == 0) return hipSuccess;
  if (dp < dim || (dp != dim && !featsP)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_feats_kernel, dim3((unsigned)((n_frames + 63) / 64)), dim3(256), 0, stream, feats, n_frames, dim, dp, ldT, out,
                     dp != dim ? featsP : nullptr);
  return hipGetLastError();
}

int gmm_refine_slots(uint32_t max_dens, uint32_t padded_dim) { return padded_dim > 39 ? 32 : max_dens <= 8 ? 8 : max_dens <= 16 ? 16 : 32; }

// The padded (odd) dimension a model of dimension `dim` is refined in; 0: beyond the refinement's instantiations (dim > 62).
// 25 and 39 are the reference's real-speech and the benchmark's dimensions and stay exact fits.
uint32_t gmm_refine_padded_dim(uint32_t dim) {
  static const uint32_t bounds[] = {9, 17, 25, 33, 39, 47, 55, 63};
  // an odd dim needs its pairs (dim - 1) and the tail: bound >= dim; an even dim needs its pairs (dim) below the tail: bound >= dim + 1
  const uint32_t need = dim | 1u;
  for (uint32_t b : bounds)
    if (need <= b) return b;
  return 0;
}

static int refine_threads(uint32_t dp) { return dp <= 47 ? kRThreads : kRThreadsWide; }
static int refine_spw(const GmmRefineArgs& a) {
  // 8 states per workgroup when their parameters fit the 160 KiB of LDS (dim <= 39 at 32 slots), else 4
  return (size_t)(2 * a.dim + 2) * a.n_slots * 8 * 8 <= 160 * 1024 ? 8 : 4;
}
static void refine_grid(const GmmRefineArgs& a, int spw, uint32_t* n_sgroups, uint64_t* splits, uint64_t* frames_per_split) {
  const uint64_t nt = (uint64_t)refine_threads(a.dim);
  *n_sgroups = (a.n_pstates + spw - 1) / spw;
  // two rounds of workgroups over the 256 CUs at least (one workgroup per CU when it takes the whole LDS), without
  // cutting the frame range below one pass of the threads
  uint64_t sp = std::max<uint64_t>(1, (512 + *n_sgroups - 1) / *n_sgroups);
  sp = std::min<uint64_t>(sp, (a.n_frames + nt - 1) / nt);
  sp = std::max<uint64_t>(sp, 1);
  *frames_per_split = (a.n_frames + sp - 1) / sp;
  *splits = *frames_per_split ? (a.n_frames + *frames_per_split - 1) / *frames_per_split : 1;
}
size_t gmm_refine_ring_words(const GmmRefineArgs& a) {
  uint32_t g; uint64_t sp, fps;
  refine_grid(a, refine_spw(a), &g, &sp, &fps);
  return (size_t)g * sp * kRWaves * kRingWave;
}

// how many entries a wave's segment of one panel holds: a quarter of the wave's (frame, panel) pairs (configs[4]: 5 % of them are listed)
static uint32_t defer_cap_for(const GmmRefineArgs& a, int spw) {
  uint32_t g; uint64_t sp, fps;
  refine_grid(a, spw, &g, &sp, &fps);
  const uint64_t per_wave = ((fps + kRThreads - 1) / kRThreads) * 64;  // frames one wave sees
  return (uint32_t)std::min<uint64_t>(((per_wave / 4 + 63) & ~(uint64_t)63) + 64, 1u << 30);
}
void gmm_refine_defer_layout(const GmmRefineArgs& a, size_t budget_bytes, uint32_t* cap, size_t* n_entries, size_t* n_counts) {
  *cap = 0; *n_entries = 0; *n_counts = 0;
  if (a.chunks < 2 || a.dim > 39 || a.n_slots != 32 || a.n_frames == 0) return;  // (instantiated for the 768-thread, 8-panel geometry)
  uint32_t g; uint64_t sp, fps;
  refine_grid(a, 8, &g, &sp, &fps);
  uint32_t c = defer_cap_for(a, 8);
  if (const char* e = getenv("SRGPU_DEFER_CAP")) c = std::max(1u, std::min(c, (uint32_t)strtoul(e, nullptr, 10)));  // (tests: full segments)
  const size_t segs = (size_t)g * sp * kRWaves * 8;
  if (segs * c * 16 > budget_bytes) return;
  *cap = c; *n_entries = segs * c; *n_counts = segs;
}

template <int DT, int NS, int SPW, int CH, int NT, int VS = 1>
static hipError_t launch_refine_one(const GmmRefineArgs& a0, hipStream_t stream) {
  GmmRefineArgs a = a0;
  const size_t state_bytes = (size_t)(2 * DT + 2) * NS * 8;
  const size_t smem = (SPW * state_bytes + 1023) & ~(size_t)1023;
  uint32_t n_sgroups;
  uint64_t splits;
  refine_grid(a, SPW, &n_sgroups, &splits, &a.frames_per_split);
  const dim3 grid(n_sgroups, (unsigned)splits), block(NT);
  if constexpr (CH > 1 && NT == kRThreads && SPW == 8) {
    if (a.defer && a.defer_cnt && a.defer_cap) {  // main pass with deferred lists, then every wave on the leftovers
      auto k1 = gmm_refine_kernel<DT, NS, SPW, CH, NT, VS, true>;
      auto k2 = gmm_drain_kernel<DT, NS, SPW, CH, NT, VS>;
      hipError_t e = hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k1, grid, block, smem, stream, a);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      hipLaunchKernelGGL(k2, grid, block, smem, stream, a);
      return hipGetLastError();
    }
  }
  a.defer = nullptr; a.defer_cnt = nullptr; a.defer_cap = 0;
  auto kernel = gmm_refine_kernel<DT, NS, SPW, CH, NT, VS>;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, grid, block, smem, stream, a);
  return hipGetLastError();
}

// padded dimension <= 39: 768 threads, 8 states per workgroup, panels of 8 / 16 / 32 slots
template <int DT>
static hipError_t launch_refine_narrow(const GmmRefineArgs& a, hipStream_t stream) {
  switch (a.n_slots) {
    case 8: return launch_refine_one<DT, 8, 8, 1, kRThreads>(a, stream);
    case 16: return launch_refine_one<DT, 16, 8, 1, kRThreads>(a, stream);
    case 32:  // (a mixture of more than 32 densities always has 32-slot panels)
      if (a.chunks == 2) return launch_refine_one<DT, 32, 8, 2, kRThreads>(a, stream);
      if (a.chunks == 4) return launch_refine_one<DT, 32, 8, 4, kRThreads>(a, stream);
      if (a.chunks == 8) return launch_refine_one<DT, 32, 8, 4, kRThreads, 2>(a, stream);  // 129 .. 256 densities: two halves of four chunks
      return launch_refine_one<DT, 32, 8, 1, kRThreads>(a, stream);
    default: return hipErrorInvalidValue;
  }
}
// padded dimension 55 / 63: 512 threads (2 waves per SIMD: the features alone take up to 126 registers), 4 states per workgroup
// (8 would not fit the LDS at 32 slots), 32-slot panels whatever the model's largest mixture
template <int DT>
static hipError_t launch_refine_wide(const GmmRefineArgs& a, hipStream_t stream) {
  if (a.n_slots != 32) return hipErrorInvalidValue;
  if (a.chunks == 2) return launch_refine_one<DT, 32, 4, 2, kRThreadsWide>(a, stream);
  if (a.chunks == 4) return launch_refine_one<DT, 32, 4, 4, kRThreadsWide>(a, stream);
  return launch_refine_one<DT, 32, 4, 1, kRThreadsWide>(a, stream);
}

// a.dim = the PADDED dimension (gmm_refine_padded_dim); a.feats rows of a.dim floats, a.featsT a.dim rows, a.rows 2 a.dim + 2 planes
hipError_t launch_gmm_refine(const GmmRefineArgs& a, hipStream_t stream) {
  if (a.n_frames == 0) return hipSuccess;
  if (!a.ring || (uint64_t)a.dim * a.n_frames_ld * 4u >= (1ull << 32)) return hipErrorInvalidValue;  // (buffer offsets are 32 bit)
  if (a.chunks != 1 && (a.n_slots != 32 || (a.chunks != 2 && a.chunks != 4 && a.chunks != 8))) return hipErrorInvalidValue;
  if (a.chunks == 8 && a.dim > 39) return hipErrorInvalidValue;  // (eight panels of one state in one workgroup's LDS: padded dimension <= 39)
  switch (a.dim) {
    case 9: return launch_refine_narrow<9>(a, stream);
    case 17: return launch_refine_narrow<17>(a, stream);
    case 25: return launch_refine_narrow<25>(a, stream);
    case 33: return launch_refine_narrow<33>(a, stream);
    case 39: return launch_refine_narrow<39>(a, stream);
    case 47:  // 161 registers: still three waves per SIMD (768 threads), but 4 states per workgroup (8 do not fit the LDS)
      if (a.n_slots != 32) return hipErrorInvalidValue;
      return a.chunks == 2 ? launch_refine_one<47, 32, 4, 2, kRThreads>(a, stream) : a.chunks == 4 ? launch_refine_one<47, 32, 4, 4, kRThreads>(a, stream)
                           : launch_refine_one<47, 32, 4, 1, kRThreads>(a, stream);
    case 55: return launch_refine_wide<55>(a, stream);
    case 63: return launch_refine_wide<63>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace srgpu
