// mfma_f64_rate.hip -- micro-benchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
// The MI355X guide has no FP64 row; this pins the denominator used for roofline.frac.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
  double acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void muladd_loop(double* out, int iters, double a0, double b0) {
  double acc[8];
  for (int i = 0; i < 8; i++) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = acc[i] * a; acc[i] = acc[i] + b; }
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();  // warm
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * cus * 8 * 256);
  const int iters = 20000;
  for (int wg_per_cu : {1, 2}) {
    const int grid = cus * wg_per_cu;
    double ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 2048.0;
    printf("mfma_f64_16x16x4 x4acc  %d wg/cu: %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD at 2.4GHz)\n", wg_per_cu, ms,
           flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * wg_per_cu));
    ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    flops = (double)grid * 4 * iters * 1 * 2048.0;
    printf("mfma_f64_16x16x4 x1acc  %d wg/cu: %.3f ms  %.2f TFLOP/s  (dependent chain)\n", wg_per_cu, ms, flops / ms * 1e-9);
  }
  for (int wg_per_cu : {4, 8}) {
    const int grid = cus * wg_per_cu;
    double ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    double flops = (double)grid * 256 * iters * 8 * 2.0;
    printf("v_fma_f64   %d wg/cu: %.3f ms  %.2f TFLOP/s\n", wg_per_cu, ms, flops / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(muladd_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("v_mul_f64+v_add_f64 %d wg/cu: %.3f ms  %.2f Tops/s (unfused ops)\n", wg_per_cu, ms, flops / ms * 1e-9);
  }
  hipFree(out);
  return 0;
}
