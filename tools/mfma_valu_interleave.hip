// Microbenchmark 2: ONE wave interleaving its own vector work between its MFMAs.
// Every wave runs `reps` stages of 40 v_mfma_f32_32x32x16_f16 (two accumulator chains) with V vector instructions of the mask
// epilogue's mix (v_min3 / v_cmp_ngt + v_addc, on registers the MFMAs do not touch) after each MFMA; 1 or 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_interleave.hip -o tools/mfma_valu_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define VEC2(m, lim, a, b)                                                                                     \
  asm volatile("v_cmp_ngt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(lim), "v"(a) : "vcc"); \
  (void)(b)

template <int V, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
  v16f acc0, acc1;
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float v[16];
  for (int j = 0; j < 16; j++) v[j] = 0.37f * (float)((lane * 7 + j * 13) % 97);
  uint32_t mask = 0;
  float limit = 17.0f;
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int s = 0; s < 40; s++) {
      if (MFMA) {
        if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < V / 2; q++)
        asm volatile("v_cmp_ngt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(limit), "v"(v[(s * 5 + q) & 15]) : "vcc");
    }
    limit += 1e-6f;
  }
  float sink = (float)mask;
  for (int r = 0; r < 16; r++) sink += acc0[r] + acc1[r];
  out[blockIdx.x * 512 + threadIdx.x] = sink;
}

template <int V, bool MFMA>
static void run(float* d, int reps, int threads, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, MFMA>), dim3(256), dim3(threads), 0, 0, d, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%d wave(s)/SIMD, %s, %2d vector instructions after each: %.3f ms for %d stages -> %.0f ns per stage\n", threads / 256, what, V, best, reps, best * 1e6 / reps);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  for (int threads = 256; threads <= 512; threads += 256) {
    run<0, true>(d, reps, threads, "40 MFMAs per stage");
    run<10, false>(d, reps, threads, "no MFMAs (40 slots)");
    run<2, true>(d, reps, threads, "40 MFMAs per stage");
    run<4, true>(d, reps, threads, "40 MFMAs per stage");
    run<6, true>(d, reps, threads, "40 MFMAs per stage");
    run<8, true>(d, reps, threads, "40 MFMAs per stage");
    run<10, true>(d, reps, threads, "40 MFMAs per stage");
    run<12, true>(d, reps, threads, "40 MFMAs per stage");
    run<16, true>(d, reps, threads, "40 MFMAs per stage");
  }
  return 0;
}
