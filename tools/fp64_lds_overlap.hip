// Microbenchmark: do ds_read_b64 returns cost the FP64 vector pipe issue time?  (gmm_refine_kernel: one evaluation = 160 FP64
// instructions + 80 ds_read_b64; each alone runs at its rate, together they take ~ the sum of FP64 quads + return quads.)
// 256 workgroups of 768 threads (3 waves per SIMD), each wave iterates: NF independent-chain FP64 operations and NL ds_read_b64
// of conflict-free addresses whose results feed the chains.   build: hipcc --offload-arch=gfx950 -O3 tools/fp64_lds_overlap.hip -o tools/fp64_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NF, int NL>
__global__ __launch_bounds__(768) void k(double* out, int reps) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 768) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const volatile __attribute__((address_space(3))) double* col =
      (const volatile __attribute__((address_space(3))) double*)(lds + ((lane * 7) & 31));  // 32 slots of 8 bytes: conflict-free
  double c[8];
  for (int j = 0; j < 8; j++) c[j] = 1.0 + 0.001 * (lane + j);
  for (int it = 0; it < reps; it++) {
    double p[NL > 0 ? NL : 1];
#pragma unroll
    for (int j = 0; j < NL; j++) p[j] = col[(j % 64) * 32];
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const double q = NL > 0 ? p[j % (NL > 0 ? NL : 1)] : 1.0000001;
      if (j & 1) c[j & 7] = c[j & 7] * q; else c[j & 7] = c[j & 7] + q;
    }
    if (NF == 0) {
#pragma unroll
      for (int j = 0; j < NL; j++) asm volatile("" :: "v"(p[j]));
    }
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += c[j];
  out[blockIdx.x * 768 + tid] = s;
}

template <int NF, int NL>
static void run(double* d, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, NL>), dim3(256), dim3(768), 4096 * 8, 0, d, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%3d FP64 operations + %3d ds_read_b64 per wave and iteration: %.0f ns per iteration (3 waves per SIMD)\n", NF, NL, best * 1e6 / reps);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20000;
  double* d;
  (void)hipMalloc(&d, 256 * 768 * 8);
  run<160, 0>(d, reps);
  run<0, 80>(d, reps);
  run<160, 20>(d, reps);
  run<160, 40>(d, reps);
  run<160, 80>(d, reps);
  run<160, 120>(d, reps);
  run<80, 80>(d, reps);
  return 0;
}
