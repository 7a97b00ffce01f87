// Cycles per instruction of the fp16 MFMA shapes a K = 80 prefilter would mix: v_mfma_f32_16x16x32_f16 (gfx950's
// double-K form) against the older v_mfma_f32_16x16x16_f16, back to back on one SIMD, one wave per SIMD, s_memtime around
// 4 x 256 instructions on four independent accumulators; and the k-step pattern 32 + 32 + 16 against 32 + 32 + 32.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_shapes.hip -o tools/mfma_f16_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: 16x16x32 only, 1: 16x16x16 only, 2: 32+32+16, 3: 32+32+32
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  f16x8 a8, b8;
  f16x4 a4, b4;
  for (int j = 0; j < 8; j++) { a8[j] = (_Float16)(0.001f * (lane + j)); b8[j] = (_Float16)(0.002f * (lane - j)); }
  for (int j = 0; j < 4; j++) { a4[j] = a8[j]; b4[j] = b8[j]; }
  v4f acc[4];
  for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int s = 0; s < 48; s++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const bool small = MODE == 1 || (MODE == 2 && s % 3 == 2);
        if (small) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sink = 0.f;
  for (int i = 0; i < 4; i++) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sink;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(float* d, unsigned long long* c, const char* what) {
  const int reps = 200;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  unsigned long long cyc = 0;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, d, c, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    (void)hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
  }
  const double n = 192.0 * reps;
  printf("%-34s %.3f ms, %.1f ns per MFMA, s_memtime ticks per MFMA %.2f (100 MHz ticks x clock ratio unknown: compare rows)\n", what, best, best * 1e6 / n, (double)cyc / n);
}

int main() {
  float* d; unsigned long long* c;
  (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 8);
  run<0>(d, c, "16x16x32_f16 only");
  run<1>(d, c, "16x16x16_f16 only");
  run<3>(d, c, "k-steps 32 + 32 + 32 (K = 96)");
  run<2>(d, c, "k-steps 32 + 32 + 16 (K = 80)");
  return 0;
}
