// mfma_f64_rate.hip -- micro-benchmark: sustained FP64 rates on gfx950 (MI355X).
//   v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 at 1..8 waves per SIMD, v_fma_f64 at 1..8 waves per
//   SIMD, unfused mul+add, and MFMA waves co-resident with FMA waves (do the pipes add up?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
// The MI355X guide has no FP64 row; this pins the denominators used in DESIGN.md / bench.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode: 0 = mfma 16x16x4 (4 acc), 1 = fma (8 acc), 2 = mixed: waves with (wave&1)==0 do mfma, others fma,
//       3 = mfma 4x4x4 (4 acc), 4 = mul+add unfused, 5 = mfma 16x16x4 single accumulator chain
template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(double* out, int iters, double a0, double b0, unsigned long long* clk) {
  const int wave = threadIdx.x >> 6;
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-10;
  double s = 0;
  unsigned long long t0 = 0, r0 = 0;
  if (clk && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 5 || (MODE == 2 && (wave & 1) == 0);
  if (do_mfma) {
    v4d acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (v4d){0, 0, 0, 0};
    double m = 0;
    for (int it = 0; it < iters; it++) {
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; i++) m = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m, 0, 0, 0);
      } else if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += m;
  } else if (MODE == 4) {
#pragma clang fp contract(off)
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) { acc[i] = acc[i] * a; acc[i] = acc[i] + b; }
    }
    for (int i = 0; i < 8; i++) s += acc[i];
  } else {
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a, b);
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    for (int i = 0; i < 8; i++) s += acc[i];
  }
  if (clk && threadIdx.x == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int grid, int iters, double* out, unsigned long long* clk, int reps, double* ghz) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, clk);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * grid);
  (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < grid; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  *ghz = cyc / real * 0.1;  // s_memrealtime ticks at 100 MHz
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  double* out; (void)hipMalloc(&out, sizeof(double) * cus * 8 * 256);
  unsigned long long* clk; (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 8);
  const int iters = 40000, reps = 5;
  double ghz;
  for (int w : {1, 2, 4, 8}) {
    const int grid = cus * w;
    double ms = run<0>(grid, iters, out, clk, reps, &ghz);
    double fl = (double)grid * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 4acc %d waves/SIMD: %8.3f ms %6.2f TFLOP/s  clock %.2f GHz  %.1f cyc per MFMA per SIMD\n", w, ms,
           fl / ms * 1e-9, ghz, ms * 1e-3 * ghz * 1e9 / (iters * 4.0 * w));
  }
  for (int w : {1, 2}) {
    const int grid = cus * w;
    double ms = run<5>(grid, iters, out, clk, reps, &ghz);
    double fl = (double)grid * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 1acc %d waves/SIMD: %8.3f ms %6.2f TFLOP/s  clock %.2f GHz\n", w, ms, fl / ms * 1e-9, ghz);
  }
  for (int w : {1, 2, 4, 8}) {
    const int grid = cus * w;
    double ms = run<3>(grid, iters, out, clk, reps, &ghz);
    double fl = (double)grid * 4 * iters * 4 * 512.0;
    printf("mfma_f64_4x4x4_4b     %d waves/SIMD: %8.3f ms %6.2f TFLOP/s  clock %.2f GHz\n", w, ms, fl / ms * 1e-9, ghz);
  }
  for (int w : {1, 2, 4, 8}) {
    const int grid = cus * w;
    double ms = run<1>(grid, iters, out, clk, reps, &ghz);
    double fl = (double)grid * 256 * iters * 16 * 2.0;
    printf("v_fma_f64             %d waves/SIMD: %8.3f ms %6.2f TFLOP/s  clock %.2f GHz\n", w, ms, fl / ms * 1e-9, ghz);
  }
  for (int w : {2, 4, 8}) {
    const int grid = cus * w;
    double ms = run<4>(grid, iters, out, clk, reps, &ghz);
    double ops = (double)grid * 256 * iters * 16.0;
    printf("v_mul_f64 + v_add_f64 %d waves/SIMD: %8.3f ms %6.2f Tops/s   clock %.2f GHz\n", w, ms, ops / ms * 1e-9, ghz);
  }
  for (int w : {1, 2, 4}) {  // w workgroups per CU, each 2 MFMA waves + 2 FMA waves
    const int grid = cus * w;
    double ms = run<2>(grid, iters, out, clk, reps, &ghz);
    double fl_m = (double)grid * 2 * iters * 4 * 2048.0, fl_v = (double)grid * 128 * iters * 16 * 2.0;
    printf("mixed (2 mfma + 2 fma waves per WG) %d WG/CU: %8.3f ms  mfma %6.2f + fma %6.2f = %6.2f TFLOP/s  clock %.2f GHz\n", w, ms,
           fl_m / ms * 1e-9, fl_v / ms * 1e-9, (fl_m + fl_v) / ms * 1e-9, ghz);
  }
  (void)hipFree(out); (void)hipFree(clk);
  return 0;
}
