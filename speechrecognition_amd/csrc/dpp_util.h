// dpp_util.h -- wave and row reductions with DPP, v_min_f64, LDS ds_min_f64: shared by the search kernels
// (viterbi_fast.hip, viterbi_words.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srgpu {

static constexpr double kInfF = __builtin_huge_val();

// ---- DPP helpers --------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);  // lanes without a source keep v
}
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_d(double v) {
  return __hiloint2double(dpp_i<CTRL, ROW_MASK>(__double2hiint(v)), dpp_i<CTRL, ROW_MASK>(__double2loint(v)));
}
__device__ inline double readlane63_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// v_min_f64 directly: one instruction per reduction step instead of compare + two selects (fmin() would canonicalise
// its operands first).  Hypothesis scores are never NaN; +inf is an ordinary operand.
__device__ inline double dmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// LDS ds_min_f64 without a return value, then the wait that has to stand between such an atomic and the barrier behind which
// the cell is read -- ONE asm statement, so that no call site can have the first without the second.  The compiler does not see a
// DS instruction inside an asm, so its own "s_waitcnt lgkmcnt(0)" before s_barrier is there only when some OTHER LDS access
// happens to be pending; without the wait a wave can pass the barrier while its minimum is still queued, and the waves that read
// the cell first see different minima (round 3: one traceback entry in ~1e5 wrong, run to run, once a second copy of the frame
// loop was compiled without it; until round 4 the wait was a separate helper every call site had to remember).
// tests/test_isa_cpu.py checks the binary: every ds_min_f64 is followed by s_waitcnt lgkmcnt(0) before the next s_barrier.
__device__ inline uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ inline void publish_min_f64_lds(double* cell, double v) {
  asm volatile("ds_min_f64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(lds_addr(cell)), "v"(v) : "memory");
}
// Five LDS ds_min_u32 without return values in one statement, then their wait: cell0 and the four consecutive cells at cells4,
// a value of 0xFFFFFFFF where a lane has nothing to contribute (the minimum with it is a no-op).  (atomicMin() on an LDS cell goes
// through the compiler's atomic optimiser: a scalar loop over the active lanes -- s_ff1, v_readlane, s_min per lane -- before ONE
// atomic; the search kernels call it from the one or two lanes that hold a frame's minimum, five times in a row, on the wave every
// other wave of the workgroup then waits for at the barrier: 125 of that wave's ~650 instructions per frame, profiles/r4_words_stamps.txt.)
// The cells are read after a later workgroup barrier, and the compiler's own s_waitcnt before a barrier does not cover an asm: the
// wait is part of the statement, as above.
__device__ inline void lds_min5_u32(uint32_t* cell0, uint32_t v0, uint32_t* cells4, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4) {
  asm volatile("ds_min_u32 %0, %1\n\tds_min_u32 %2, %3\n\tds_min_u32 %2, %4 offset:4\n\tds_min_u32 %2, %5 offset:8\n\tds_min_u32 %2, %6 offset:12\n\t"
               "s_waitcnt lgkmcnt(0)"
               : : "v"(lds_addr(cell0)), "v"(v0), "v"(lds_addr(cells4)), "v"(v1), "v"(v2), "v"(v3), "v"(v4) : "memory");
}
// two cells, both atomics in flight together, one wait
__device__ inline void publish_min2_f64_lds(double* cell0, double v0, double* cell1, double v1) {
  asm volatile("ds_min_f64 %0, %1\n\tds_min_f64 %2, %3\n\ts_waitcnt lgkmcnt(0)"
               : : "v"(lds_addr(cell0)), "v"(v0), "v"(lds_addr(cell1)), "v"(v1) : "memory");
}
// full-wave minimum, returned to every lane
__device__ inline double wave_min_dpp(double v) {
  v = dmin(v, dpp_d<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
  v = dmin(v, dpp_d<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
  v = dmin(v, dpp_d<0x141, 0xF>(v));   // row_half_mirror
  v = dmin(v, dpp_d<0x140, 0xF>(v));   // row_mirror
  v = dmin(v, dpp_d<0x142, 0xA>(v));   // row_bcast15 -> rows 1, 3
  v = dmin(v, dpp_d<0x143, 0xC>(v));   // row_bcast31 -> rows 2, 3
  return readlane63_d(v);
}
__device__ inline uint32_t wave_min_u32_dpp(uint32_t v) {
  v = umin(v, (uint32_t)dpp_i<0xB1, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x4E, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x141, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x140, 0xF>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x142, 0xA>((int)v));
  v = umin(v, (uint32_t)dpp_i<0x143, 0xC>((int)v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// full-wave lexicographic (value, index) minimum, returned to every lane: the minimum value first, then the smallest
// index among the lanes that hold it
__device__ inline void wave_min_idx_dpp(double& v, uint32_t& idx) {
  const double m = wave_min_dpp(v);
  idx = wave_min_u32_dpp(v == m ? idx : 0xFFFFFFFFu);
  v = m;
}
// the same over a row of 16 lanes, result in every lane of the row (the cross-wave stage: each row holds all <= 16 partials)
__device__ inline double row_min_dpp(double v) {
  v = dmin(v, dpp_d<0xB1, 0xF>(v));
  v = dmin(v, dpp_d<0x4E, 0xF>(v));
  v = dmin(v, dpp_d<0x141, 0xF>(v));
  v = dmin(v, dpp_d<0x140, 0xF>(v));
  return v;
}
__device__ inline void row_min_idx_dpp(double& v, uint32_t& idx) {
  const double m = row_min_dpp(v);
  uint32_t c = v == m ? idx : 0xFFFFFFFFu;
  c = umin(c, (uint32_t)dpp_i<0xB1, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x4E, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x141, 0xF>((int)c));
  c = umin(c, (uint32_t)dpp_i<0x140, 0xF>((int)c));
  idx = c;
  v = m;
}

}  // namespace srgpu
