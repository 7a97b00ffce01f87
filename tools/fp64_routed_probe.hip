// Microbenchmark (round 4, VERDICT r3 "Next" #4a): what would the ROUTED orientation of the refinement cost per evaluation?
// Today a lane owns a FRAME (features in registers) and reads its candidate's 80 parameters from LDS: 160 FP64 operations +
// 80 ds_read_b64 = 640 bytes per lane through the register write port (tools/fp64_lds_width.hip: 1433 ns per iteration at three
// waves per SIMD).  Routed: a lane owns a DENSITY (parameters resident in 160 registers), pairs are routed to it and the lane
// gathers its pair's 39 features (156 bytes) from a tile of feature rows in LDS: 160 FP64 + 39 v_cvt_f64_f32 + 39 ds_read_b32 from
// a RANDOM row per lane (row stride 39 floats: the bank pattern a gather has).  Register budget: 160 parameters + 78 features in
// flight -> at most 2 waves per SIMD (512 threads per workgroup here), 1 with deeper pipelining (256).
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_routed_probe.hip -o tools/fp64_routed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int THREADS, bool GATHER, bool ARITH, bool B128>
__global__ __launch_bounds__(THREADS) void k(double* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) float rows[];  // 768 rows x 40 floats (39 features + pad: 160 bytes)
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int STRIDE = B128 ? 40 : 39;
  for (int i = tid; i < 768 * 40; i += THREADS) rows[i] = 1.0f + 1e-4f * (i % 977);
  __syncthreads();
  double c[8];
  for (int j = 0; j < 8; j++) c[j] = 1.0 + 0.001 * (lane + j);
  unsigned h = tid * 2654435761u;
  for (int it = 0; it < reps; it++) {
    h = h * 1664525u + 1013904223u;
    const unsigned row = (h >> 10) % 768u;  // a different, unrelated row per lane and iteration
    double x[40];
    if (GATHER) {
      if (B128) {
        const volatile __attribute__((address_space(3))) f4* p = (const volatile __attribute__((address_space(3))) f4*)(rows + row * STRIDE);
#pragma unroll
        for (int j = 0; j < 10; j++) { const f4 v = p[j]; x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w; }
      } else {
        const volatile __attribute__((address_space(3))) float* p = (const volatile __attribute__((address_space(3))) float*)(rows + row * STRIDE);
#pragma unroll
        for (int j = 0; j < 39; j++) x[j] = (double)p[j];
        x[39] = 1.0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 40; j++) x[j] = 1.0000001 + 1e-9 * j;
    }
    if (ARITH) {
#pragma unroll
      for (int j = 0; j < 160; j++) {
        const double q = x[j % 40];
        if (j & 1) c[j & 7] = c[j & 7] * q; else c[j & 7] = c[j & 7] + q;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 40; j++) asm volatile("" :: "v"(x[j]));
    }
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += c[j];
  out[blockIdx.x * THREADS + tid] = s;
}

template <int THREADS, bool GATHER, bool ARITH, bool B128>
static void run(double* d, int reps, const char* what) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<THREADS, GATHER, ARITH, B128>), dim3(256), dim3(THREADS), 768 * 40 * 4, 0, d, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%d wave(s) per SIMD, %-78s %6.0f ns per iteration = %6.0f ns per evaluation and SIMD\n", THREADS / 256, what, best * 1e6 / reps,
         best * 1e6 / reps / (THREADS / 256));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20000;
  double* d;
  (void)hipMalloc(&d, 256 * 768 * 8);
  run<256, false, true, false>(d, reps, "160 FP64 operations, features in registers");
  run<256, true, false, false>(d, reps, "39 ds_read_b32 from a random 156-byte row per lane + 39 v_cvt_f64_f32, no arithmetic");
  run<256, true, true, false>(d, reps, "160 FP64 + 39 ds_read_b32 (random rows) + 39 v_cvt_f64_f32");
  run<256, true, true, true>(d, reps, "160 FP64 + 10 ds_read_b128 (random 160-byte rows) + 40 v_cvt_f64_f32");
  run<512, false, true, false>(d, reps, "160 FP64 operations, features in registers");
  run<512, true, true, false>(d, reps, "160 FP64 + 39 ds_read_b32 (random rows) + 39 v_cvt_f64_f32");
  run<512, true, true, true>(d, reps, "160 FP64 + 10 ds_read_b128 (random 160-byte rows) + 40 v_cvt_f64_f32");
  run<768, false, true, false>(d, reps, "160 FP64 operations, features in registers  (today's occupancy, for scale)");
  run<768, true, true, true>(d, reps, "160 FP64 + 10 ds_read_b128 (random 160-byte rows) + 40 v_cvt_f64_f32 (registers would not allow it)");
  printf("today (tools/fp64_lds_width.hip, 3 waves per SIMD): 160 FP64 + 80 ds_read_b64 = 1433 ns per iteration = 478 ns per evaluation and SIMD\n");
  return 0;
}
