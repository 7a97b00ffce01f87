// Microbenchmark: do the prefilter's two instruction mixes overlap when they run in the two waves of a SIMD?
// 512-thread workgroups, one per CU: waves 0-3 run `reps` x 40 v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 (a stage's MFMAs),
// waves 4-7 `reps` x the mask epilogue's vector mix (16 v_min3 + 32 x (v_cmp_ngt + v_addc) per 32 values, x 4).
// mode 1: MFMA waves only, 2: vector waves only, 3: both.   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE, int PACE>
__global__ __launch_bounds__(512) void k(float* out, int reps, int mode, int prio, int swap) {
  const int wave = (threadIdx.x >> 6) & 7, lane = threadIdx.x & 63;  // swap: the vector waves are the older ones
  if (wave >= 4 && (prio & 1)) __builtin_amdgcn_s_setprio(3);
  if (wave < 4 && (prio & 2)) __builtin_amdgcn_s_setprio(3);
  float sink = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f16x8 a, b;
      for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
      if (SHAPE == 32) {
        v16f acc0, acc1;
        for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
        for (int it = 0; it < reps; it++) {
#pragma unroll
          for (int s = 0; s < 20; s++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            if (PACE == 1) asm volatile("s_nop 0");
            if (PACE == 2) asm volatile("s_nop 3");
            if (PACE == 3) asm volatile("s_nop 7");
            if (PACE == 4) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
            if (PACE == 5) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
            if (PACE == 6) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 4"); }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            if (PACE == 1) asm volatile("s_nop 0");
            if (PACE == 2) asm volatile("s_nop 3");
            if (PACE == 3) asm volatile("s_nop 7");
            if (PACE == 4) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
            if (PACE == 5) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
            if (PACE == 6) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 4"); }
          }
        }
        for (int r = 0; r < 16; r++) sink += acc0[r] + acc1[r];
      } else {
        v4f acc[4];
        for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < reps; it++) {
#pragma unroll
          for (int s = 0; s < 24; s++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
      }
    }
  } else if (mode & 2) {
    float v[32];
    for (int j = 0; j < 32; j++) v[j] = 0.37f * (float)((lane * 7 + j * 13) % 97);
    uint32_t mask = 0;
    const int vmix = prio >> 2;  // 0: the mask epilogue (v_min3 + v_cmp/v_addc), 1: 400 v_fma_f32, 2: v_sub_f32 + v_alignbit_b32 per value
    if (vmix == 1) {
      for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int u = 0; u < 12; u++)
#pragma unroll
          for (int j = 0; j < 32; j++) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
      }
      for (int j = 0; j < 32; j++) mask += (uint32_t)v[j];
    } else if (vmix == 2) {
      for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          float amin = __builtin_huge_valf();
#pragma unroll
          for (int j = 0; j < 32; j += 8)
            asm volatile("v_min3_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %3, %4\n\tv_min3_f32 %0, %0, %5, %6\n\tv_min3_f32 %0, %0, %7, %8"
                : "+v"(amin) : "v"(v[j]), "v"(v[j + 1]), "v"(v[j + 2]), "v"(v[j + 3]), "v"(v[j + 4]), "v"(v[j + 5]), "v"(v[j + 6]), "v"(v[j + 7]));
          const float limit = amin + 1.5f + (float)it * 1e-9f;
#pragma unroll
          for (int j = 0; j < 32; j++) {
            float t;
            asm volatile("v_sub_f32 %1, %2, %3\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(mask), "=&v"(t) : "v"(v[j]), "v"(limit));
          }
          v[u] += (float)(mask & 1u);
        }
      }
    } else
    for (int it = 0; it < reps; it++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {  // four (frame block, state pair) units per stage
        float amin = __builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < 32; j += 8)
          asm volatile("v_min3_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %3, %4\n\tv_min3_f32 %0, %0, %5, %6\n\tv_min3_f32 %0, %0, %7, %8"
              : "+v"(amin) : "v"(v[j]), "v"(v[j + 1]), "v"(v[j + 2]), "v"(v[j + 3]), "v"(v[j + 4]), "v"(v[j + 5]), "v"(v[j + 6]), "v"(v[j + 7]));
        const float limit = amin + 1.5f + (float)it * 1e-9f;
#pragma unroll
        for (int j = 0; j < 32; j += 8)
          asm volatile("v_cmp_ngt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %3, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %4, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %5, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %6, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %7, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %8, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
              "v_cmp_ngt_f32 vcc, %9, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
              : "+v"(mask) : "v"(limit), "v"(v[j]), "v"(v[j + 1]), "v"(v[j + 2]), "v"(v[j + 3]), "v"(v[j + 4]), "v"(v[j + 5]), "v"(v[j + 6]), "v"(v[j + 7]) : "vcc");
        v[u] += (float)(mask & 1u);
      }
    }
    sink = (float)mask + v[0];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sink;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 32; shape <= 32; shape += 16)
   for (int cfg = 0; cfg < 7; cfg++)
    for (int mode = 1; mode <= 3; mode += 2) {
      const int prio = 0, swap = cfg;
      float best = 1e9f;
      for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        switch (swap) {
          case 0: hipLaunchKernelGGL((k<32, 0>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          case 1: hipLaunchKernelGGL((k<32, 1>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          case 2: hipLaunchKernelGGL((k<32, 2>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          case 3: hipLaunchKernelGGL((k<32, 3>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          case 4: hipLaunchKernelGGL((k<32, 4>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          case 5: hipLaunchKernelGGL((k<32, 5>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
          default: hipLaunchKernelGGL((k<32, 6>), dim3(256), dim3(512), 0, 0, d, reps, mode, prio, swap); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("mfma %dx%d mode %d (%s)%s%s: %.3f ms for %d stages -> %.0f ns per stage\n", shape, shape, mode,
             mode == 1 ? "MFMA waves only" : mode == 2 ? "vector waves only" : "both",
             swap == 1 ? ", s_nop 0 after each MFMA" : swap == 2 ? ", s_nop 3" : swap == 3 ? ", s_nop 7" : swap == 4 ? ", 2 x s_nop 7" : swap == 5 ? ", 3 x s_nop 7" : swap == 6 ? ", 2 x s_nop 7 + s_nop 4" : "", "", best, reps, best * 1e6 / reps);
    }
  return 0;
}
