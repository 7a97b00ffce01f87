// Stress test (round 5, VERDICT r4 "Next" #1): the hand-off the refinement kernel's candidate lists rest on.
// gmm_refine_kernel keeps wave-private lists in global memory: a lane stores a 16-byte entry, and a little later ANOTHER lane of
// the same wave loads it back; a batch also lowers a table entry with a no-return global_atomic_min_f64 behind plain stores that
// other lanes of the wave made to the same address.  Rounds 3-4 put a hand-counted `s_waitcnt vmcnt(N)` in front of the read-back;
// round 5 orders both by wavefront-scope release / acquire fences, for which the gfx950 compiler emits no wait at all (a wave's
// vector-memory operations reach an address in issue order).  This program hammers exactly that pattern with NO wait under load:
//   every wave, per iteration: 64 lanes store entries {iteration, lane tag} at PERMUTED slots of the wave's ring (a different
//   permutation per iteration), 0..3 younger stores to an unrelated table (traffic behind the entries), fence pair, every lane
//   loads slot [lane] -- written by another lane -- and checks tag and iteration; then: plain store of a large value to a
//   per-wave table cell by lane p, atomic minimum with a smaller value from lane q != p, one read-back (waited for) per
//   iteration of the cell of the PREVIOUS iteration.
// 1024 workgroups x 8 waves, every CU busy, rings re-used every iteration (L1/L2-hot lines: the case a stale copy would show in).
// build: hipcc --offload-arch=gfx950 -O3 tools/wave_handoff_stress.hip -o tools/wave_handoff_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct __attribute__((aligned(16))) Entry { unsigned it, tag; double v; };

__global__ __launch_bounds__(512) void stress(Entry* rings, double* cells, double* sink, unsigned long long* bad, unsigned long long* bad_min, int iters) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  Entry* ring = rings + (size_t)wave * 128u;
  double* cell = cells + (size_t)wave * 2u;
  double* junk = sink + (size_t)wave * 64u * 4u;
  unsigned long long wrong = 0, wrong_min = 0;
  unsigned h = wave * 2654435761u + 12345u;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const unsigned rot = (h >> 8) & 63u, mul = ((h >> 16) & 31u) * 2u + 1u;  // slot = (lane * odd + rot) mod 64: a permutation
    const unsigned slot = (lane * mul + rot) & 63u, half = (it & 1u) * 64u;
    ring[half + slot] = Entry{(unsigned)it, lane * 7919u + (unsigned)it, (double)lane};
    const unsigned extra = (h >> 4) & 3u;  // wave-uniform: 0..3 younger stores nobody reads back
    for (unsigned e = 0; e < extra; e++) junk[e * 64u + lane] = (double)it;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("" ::: "memory");
    const Entry e = ring[half + lane];  // stored by the lane l with (l * mul + rot) mod 64 == lane
    // inverse of the permutation: l = (lane - rot) * mul^-1 mod 64; checked through the tag instead of inverting
    const unsigned src = (e.tag - (unsigned)it) / 7919u;
    if (e.it != (unsigned)it || src > 63u || ((src * mul + rot) & 63u) != lane || e.v != (double)src) wrong++;
    // store-then-atomic-minimum on one address from two different lanes
    const unsigned p = (h >> 22) & 63u, q = (p + 1u + ((h >> 28) & 7u)) & 63u;
    double* c = cell + (it & 1);
    if (lane == p) *c = 1e6 + it;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == q) (void)__builtin_amdgcn_global_atomic_fmin_f64((__attribute__((address_space(1))) double*)c, (double)it);
    if (it > 0 && lane == 0) {  // the other cell: the previous iteration's (store, minimum) pair must read as the minimum
      const double got = __hip_atomic_load(cell + ((it - 1) & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got != (double)(it - 1)) wrong_min++;
    }
  }
  if (wrong) atomicAdd(bad, wrong);
  if (wrong_min) atomicAdd(bad_min, wrong_min);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000, blocks = 1024, threads = 512;
  const size_t waves = (size_t)blocks * threads / 64;
  Entry* rings; double *cells, *sink; unsigned long long *bad, *bad_min;
  hipMalloc(&rings, waves * 128 * sizeof(Entry));
  hipMalloc(&cells, waves * 2 * sizeof(double));
  hipMalloc(&sink, waves * 256 * sizeof(double));
  hipMalloc(&bad, 8); hipMalloc(&bad_min, 8);
  hipMemset(rings, 0xFF, waves * 128 * sizeof(Entry));
  hipMemset(cells, 0, waves * 2 * sizeof(double));
  hipMemset(bad, 0, 8); hipMemset(bad_min, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(stress, dim3(blocks), dim3(threads), 0, 0, rings, cells, sink, bad, bad_min, iters);
  hipEventRecord(e1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hb, hm;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hm, bad_min, 8, hipMemcpyDeviceToHost);
  printf("wave hand-off stress: %zu waves x %d iterations = %.3g entry hand-offs, %.3g store->atomic-min pairs, %.1f ms: "
         "%llu stale entries, %llu wrong minima\n", waves, iters, (double)waves * iters * 64, (double)waves * iters, ms, hb, hm);
  return (hb || hm) ? 1 : 0;
}
