// mfma_probe.hip -- derive the lane maps of v_mfma_f64_4x4x4_4b_f64 empirically (one-hot operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {  // out[pa][pb][lane]
  const int lane = threadIdx.x;
  for (int pa = 0; pa < 64; pa++)
    for (int pb = 0; pb < 64; pb++) {
      double a = lane == pa ? 1.0 : 0.0, b = lane == pb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(pa * 64 + pb) * 64 + lane] = d;
    }
}
int main() {
  double* d; (void)hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  (void)hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  // for each A lane: which B lanes pair with it (nonzero output), and to which output lane
  for (int pa = 0; pa < 64; pa++) {
    printf("A lane %2d:", pa);
    for (int pb = 0; pb < 64; pb++)
      for (int l = 0; l < 64; l++)
        if (h[(pa * 64 + pb) * 64 + l] != 0.0) printf(" (B%d->D%d)", pb, l);
    printf("\n");
  }
  return 0;
}
