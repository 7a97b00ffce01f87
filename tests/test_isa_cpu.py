"""Properties of the gfx950 BINARY that the measured speed-ups and one correctness fix rest on, checked without a GPU
(ADVICE r3): llvm-objdump / llvm-readelf on the code objects inside speechrecognition_amd/csrc/build/*.hip.o.

  (i)   every LDS `ds_min_f64` of the search kernels is followed by `s_waitcnt lgkmcnt(0)` before the next `s_barrier`.  The
        atomics are inline asm, so the compiler's own wait before a barrier does not cover them: round 3 shipped a frame loop
        without the wait for a while and one traceback entry in ~1e5 came out wrong, run to run;
  (ii)  the headline kernels have no scratch (private segment): a spill there is a slow-down nobody would notice in a test;
  (iii) the VGPR counts stay inside the occupancy each kernel's geometry assumes.
The checker itself is tested on a throw-away kernel compiled here with the wait left out (it must be flagged)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_info  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(isa_info.LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")


@pytest.fixture(scope="module")
def objects():
    from speechrecognition_amd import build
    build.build()
    with tempfile.TemporaryDirectory() as tmp:
        out = {}
        for name in ("viterbi_words", "viterbi_fast", "gmm_prefilter", "viterbi_bigram"):
            co = isa_info.code_object(name, tmp)
            out[name] = (isa_info.kernel_metadata(co), isa_info.disassembly(co))
        yield out


def unguarded_lds_atomics(insts):
    """Indices of ds_min_f64 instructions that are NOT followed by `s_waitcnt ... lgkmcnt(0)` before control can reach a barrier
    or leave the straight line: between the atomic and its wait only further LDS atomics may stand."""
    bad = []
    for i, ins in enumerate(insts):
        if not ins.startswith("ds_min_f64"):
            continue
        ok = False
        for nxt in insts[i + 1:i + 8]:
            op = nxt.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in nxt:
                ok = True
                break
            if op == "s_waitcnt" and nxt.split()[1:] in (["0"], ["0x0"]):
                ok = True
                break
            if not op.startswith("ds_min_f64"):
                break  # anything else (a branch, a barrier, arithmetic) before the wait: the pairing is not guaranteed
        if not ok:
            bad.append(i)
    return bad


def test_every_lds_min_atomic_is_waited_for_before_the_barrier(objects):
    seen = 0
    for obj in ("viterbi_words", "viterbi_fast"):
        _, dis = objects[obj]
        for kernel, insts in dis.items():
            n = sum(1 for x in insts if x.startswith("ds_min_f64"))
            if not n:
                continue
            seen += n
            # (a one-wave workgroup, decode_fast_kernel<64, ...>, has no s_barrier at all: the wait is still required there)
            assert unguarded_lds_atomics(insts) == [], f"{kernel}: ds_min_f64 without s_waitcnt lgkmcnt(0) behind it"
    assert seen >= 2 * 18, "the search kernels publish their minima through ds_min_f64; did the kernels change?"


def test_the_checker_flags_a_kernel_without_the_wait(tmp_path):
    """What round 3's bug looked like in the binary: the atomic, then the barrier, the wait missing."""
    src = tmp_path / "racy.hip"
    src.write_text(r'''
#include <hip/hip_runtime.h>
__global__ void racy(double* out, const double* in) {
  __shared__ double cell;
  if (threadIdx.x == 0) cell = 1e300;
  __syncthreads();
  const double v = in[threadIdx.x];
  asm volatile("ds_min_f64 %0, %1" : : "v"((unsigned)(size_t)(__attribute__((address_space(3))) double*)&cell), "v"(v) : "memory");
  WAIT
  __builtin_amdgcn_s_barrier();
  out[threadIdx.x] = *(volatile double*)&cell;
}
''')
    flagged = {}
    for tag, wait in (("racy", ""), ("fixed", 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");')):
        co = tmp_path / f"{tag}.co"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "--no-gpu-bundle-output", f"-DWAIT={wait}", "-c", str(src), "-o", str(co)], check=True)
        dis = isa_info.disassembly(str(co))
        insts = next(v for k, v in dis.items() if "racy" in k)
        assert any(x.startswith("ds_min_f64") for x in insts)
        flagged[tag] = unguarded_lds_atomics(insts)
    assert flagged["racy"] and not flagged["fixed"]


# kernel (as tools/isa_info.py prints it) -> most VGPRs its launch geometry allows
VGPR_BUDGET = {
    # 768 threads = 3 waves per SIMD: 512 / 3 -> 168 (allocation granule 8)
    "gmm_refine_kernel<39, 32, 8, 1>": 168, "gmm_refine_kernel<39, 32, 8, 2>": 168, "gmm_refine_kernel<39, 32, 8, 4>": 168,
    "gmm_refine_kernel<39, 8, 8, 1>": 168, "gmm_refine_kernel<39, 16, 8, 1>": 168,
    # 256 threads, two workgroups per CU = 2 waves per SIMD
    "gmm_prefilter16_kernel<3, 4>": 256,
    # 8 waves per workgroup, two workgroups per CU = 4 waves per SIMD
    "decode_words_kernel<3, 3, false>": 128, "decode_words_kernel<1, 3, false>": 128, "decode_words_kernel<3, 4, true>": 128,
    # 1024 threads = 4 waves per SIMD (configs[4]'s lexicon: three-state rows 0 and 1, the four-state word in row 2)
    "bigram_kernel<3, 4, 3, 4>": 128, "bigram_kernel<3, 4, 3, 0>": 128,
}
NO_SCRATCH = ("gmm_refine_kernel<39, 32, 8, 1>", "gmm_refine_kernel<39, 32, 8, 2>", "gmm_refine_kernel<39, 8, 8, 1>",
              "gmm_prefilter16_kernel<3, 4>", "decode_words_kernel<3, 3, false>", "decode_words_kernel<1, 3, false>",
              "bigram_kernel<3, 4, 3, 4>", "bigram_kernel<3, 4, 3, 0>")  # (round 4: the bigram search lost its last vector spills)


def _find(objects, kernel):
    for md, _ in objects.values():
        if kernel in md:
            return md[kernel]
    raise AssertionError(f"kernel {kernel} not in the build (names: tools/isa_info.py)")


def test_headline_kernels_have_no_scratch(objects):
    for kernel in NO_SCRATCH:
        k = _find(objects, kernel)
        assert k["private_segment_fixed_size"] == 0 and k.get("vgpr_spill_count", 0) == 0, (kernel, k)


def test_vgpr_counts_fit_the_occupancy_the_kernels_are_launched_for(objects):
    for kernel, budget in VGPR_BUDGET.items():
        k = _find(objects, kernel)
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= budget, (kernel, k["vgpr_count"], budget)
