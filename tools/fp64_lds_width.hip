// Microbenchmark (round 3, follow-up of tools/fp64_lds_overlap.hip): is the vector-pipe time a returning LDS read costs paid per
// INSTRUCTION or per BYTE?  Same set-up (256 workgroups x 768 threads = 3 waves per SIMD, 160 FP64 add/mul on eight chains per
// iteration, conflict-free addresses), the same 640 bytes per lane and iteration fetched as 160 ds_read_b32, 80 ds_read_b64 or
// 40 ds_read_b128.  If gmm_refine_kernel's 80 ds_read_b64 per evaluation could become 40 ds_read_b128 at half the cost, the
// (mu, 1/var) planes would be interleaved; if the cost follows the bytes, the formulation is at its floor.
// build: hipcc --offload-arch=gfx950 -O3 tools/fp64_lds_width.hip -o tools/fp64_lds_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double2_ __attribute__((ext_vector_type(2)));

template <int NF, int NU, int W>  // NU = 8-byte units per lane and iteration, W = bytes per read instruction
__global__ __launch_bounds__(768) void k(double* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 768) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double c[8];
  for (int j = 0; j < 8; j++) c[j] = 1.0 + 0.001 * (lane + j);
  for (int it = 0; it < reps; it++) {
    double p[NU > 0 ? NU : 1];
    if (W == 8) {
      const volatile __attribute__((address_space(3))) double* col = (const volatile __attribute__((address_space(3))) double*)(lds + ((lane * 7) & 31));
#pragma unroll
      for (int j = 0; j < NU; j++) p[j] = col[(j % 64) * 32];
    } else if (W == 16) {
      const volatile __attribute__((address_space(3))) double2_* col = (const volatile __attribute__((address_space(3))) double2_*)(lds + 2 * ((lane * 7) & 31));
#pragma unroll
      for (int j = 0; j < NU / 2; j++) { const double2_ v = col[(j % 64) * 32]; p[2 * j] = v.x; p[2 * j + 1] = v.y; }
    } else {
      const volatile __attribute__((address_space(3))) int* col = (const volatile __attribute__((address_space(3))) int*)((int*)lds + ((lane * 7) & 63));
#pragma unroll
      for (int j = 0; j < NU; j++) { const int lo = col[((2 * j) % 64) * 64], hi = col[((2 * j + 1) % 64) * 64]; p[j] = __hiloint2double(hi & 0x000FFFFF | 0x3FF00000, lo); }
    }
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const double q = NU > 0 ? p[j % (NU > 0 ? NU : 1)] : 1.0000001;
      if (j & 1) c[j & 7] = c[j & 7] * q; else c[j & 7] = c[j & 7] + q;
    }
    if (NF == 0) {
#pragma unroll
      for (int j = 0; j < NU; j++) asm volatile("" :: "v"(p[j]));
    }
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += c[j];
  out[blockIdx.x * 768 + tid] = s;
}

template <int NF, int NU, int W>
static void run(double* d, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, NU, W>), dim3(256), dim3(768), 8192 * 8, 0, d, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%3d FP64 operations + %3d x ds_read_b%-3d (%4d bytes per lane) per wave and iteration: %.0f ns per iteration\n", NF, NU * 8 / W, W * 8, NU * 8,
         best * 1e6 / reps);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20000;
  double* d;
  (void)hipMalloc(&d, 256 * 768 * 8);
  run<160, 0, 8>(d, reps);
  run<0, 80, 4>(d, reps);
  run<0, 80, 8>(d, reps);
  run<0, 80, 16>(d, reps);
  run<160, 80, 4>(d, reps);
  run<160, 80, 8>(d, reps);
  run<160, 80, 16>(d, reps);
  run<160, 40, 8>(d, reps);
  run<160, 40, 16>(d, reps);
  return 0;
}
