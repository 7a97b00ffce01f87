"""Properties of the gfx950 BINARY that the measured speed-ups and one correctness fix rest on, checked without a GPU
(ADVICE r3): llvm-objdump / llvm-readelf on the code objects inside speechrecognition_amd/csrc/build/*.hip.o.

  (i)   every LDS `ds_min_f64` of the search kernels is followed by `s_waitcnt lgkmcnt(0)` before the next `s_barrier`.  The
        atomics are inline asm, so the compiler's own wait before a barrier does not cover them: round 3 shipped a frame loop
        without the wait for a while and one traceback entry in ~1e5 came out wrong, run to run;
  (ii)  the headline kernels have no scratch (private segment): a spill there is a slow-down nobody would notice in a test;
  (iii) the VGPR counts stay inside the occupancy each kernel's geometry assumes;
  (iv)  (round 5) the refinement kernel's list hand-off between the lanes of one wave -- entry stores, read back by other lanes;
        a no-return atomic minimum behind other lanes' stores to the same address -- is ordered by wavefront-scope fences, not by
        a hand-counted `s_waitcnt vmcnt(N)` as in rounds 3-4.  What the fences lower to on gfx950 (nothing: a wave's vector-memory
        operations reach an address in issue order) is pinned here on a throw-away kernel, with the agent-scope form as the
        control that the check can see a wait / write-back when one is needed; tools/wave_handoff_stress.hip hammers the pattern
        on the device (tests/test_gpu_parity.py::test_wave_handoff_stress).
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


# the LDS atomics that viterbi_words / viterbi_fast issue from INLINE ASM (dpp_util.h: publish_min*_f64_lds, lds_min5_u32).  Atomics the
# compiler emits itself (atomicAdd / atomicOr / atomicMin on __shared__, as in viterbi_bigram: ds_add_u32, ds_or_b32, ...) are seen by
# its own wait-count insertion and need no check; neither of the two objects below has a compiler-generated ds_min_*.
LDS_ATOMICS = ("ds_min_f64", "ds_min_u32")


def _is_lds_atomic(ins):
    """A no-return LDS atomic (the `_rtn_` forms hand their result to a register the compiler tracks, wait included)."""
    op = ins.split()[0]
    return "_rtn_" not in op and op.startswith(LDS_ATOMICS)


def unguarded_lds_atomics(insts):
    """Indices of no-return LDS atomics (ds_min_f64 and, since round 5, ds_min_u32 & co.: lds_min5_u32 in dpp_util.h) that are NOT
    followed by `s_waitcnt ... lgkmcnt(0)` before control can reach a barrier or leave the straight line: between the atomic and its
    wait only further LDS atomics may stand."""
    bad = []
    for i, ins in enumerate(insts):
        if not _is_lds_atomic(ins):
            continue
        ok = False
        for nxt in insts[i + 1:i + 12]:
            op = nxt.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in nxt:
                ok = True
                break
            if op == "s_waitcnt" and nxt.split()[1:] in (["0"], ["0x0"]):
                ok = True
                break
            if not _is_lds_atomic(nxt):
                break  # anything else (a branch, a barrier, arithmetic) before the wait: the pairing is not guaranteed
        if not ok:
            bad.append(i)
    return bad


def test_every_lds_min_atomic_is_waited_for_before_the_barrier(objects):
    seen = {"ds_min_f64": 0, "ds_min_u32": 0}
    for obj in ("viterbi_words", "viterbi_fast"):
        _, dis = objects[obj]
        for kernel, insts in dis.items():
            hits = [x.split()[0] for x in insts if _is_lds_atomic(x)]
            if not hits:
                continue
            for h in hits:
                seen[h] = seen.get(h, 0) + 1
            # (a one-wave workgroup, decode_fast_kernel<64, ...>, has no s_barrier at all: the wait is still required there)
            assert unguarded_lds_atomics(insts) == [], f"{kernel}: LDS atomic without s_waitcnt lgkmcnt(0) behind it"
    assert seen["ds_min_f64"] >= 2 * 18, "the search kernels publish their minima through ds_min_f64; did the kernels change?"
    assert seen["ds_min_u32"] >= 5 * 18, "lds_min5_u32 (traceback[t], e_first) should show as ds_min_u32 in the search kernels"


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
    "gmm_refine_kernel<39, 32, 8, 1, 768, 1, false>": 168, "gmm_refine_kernel<39, 32, 8, 2, 768, 1, false>": 168, "gmm_refine_kernel<39, 32, 8, 4, 768, 1, false>": 168,
    "gmm_refine_kernel<39, 8, 8, 1, 768, 1, false>": 168, "gmm_refine_kernel<39, 16, 8, 1, 768, 1, false>": 168, "gmm_refine_kernel<33, 32, 8, 1, 768, 1, false>": 168,
    "gmm_refine_kernel<47, 32, 4, 1, 768, 1, false>": 168, "gmm_refine_kernel<47, 32, 4, 2, 768, 1, false>": 168, "gmm_refine_kernel<47, 32, 4, 4, 768, 1, false>": 168,
    # padded dimension 55 / 63: 512 threads = 2 waves per SIMD
    "gmm_refine_kernel<55, 32, 4, 2, 512, 1, false>": 256, "gmm_refine_kernel<63, 32, 4, 1, 512, 1, false>": 256,
    "gmm_refine_kernel<63, 32, 4, 4, 512, 1, false>": 256, "gmm_prefilter16_kernel<4, 4>": 256,
    # mixtures of more than 32 densities: the main pass with deferred leftovers and the kernel that works them off (768 threads each)
    "gmm_refine_kernel<39, 32, 8, 2, 768, 1, true>": 168, "gmm_refine_kernel<39, 32, 8, 4, 768, 1, true>": 168, "gmm_refine_kernel<39, 32, 8, 4, 768, 2, true>": 168,
    "gmm_drain_kernel<39, 32, 8, 2, 768, 1>": 168, "gmm_drain_kernel<39, 32, 8, 4, 768, 2>": 168,
    # 256 threads, two workgroups per CU = 2 waves per SIMD
    "gmm_prefilter16_kernel<3, 4>": 256,
    # 8 waves per workgroup, two workgroups per CU = 4 waves per SIMD
    "decode_words_kernel<3, 3, false, false, 1024>": 128, "decode_words_kernel<1, 3, false, false, 1024>": 128, "decode_words_kernel<3, 4, true, false, 1024>": 128,
    # 1024 threads = 4 waves per SIMD (configs[4]'s lexicon: three-state rows 0 and 1, the four-state word in row 2)
    "bigram_kernel<3, 4, 3, 4>": 128, "bigram_kernel<3, 4, 3, 0>": 128,
}
NO_SCRATCH = ("gmm_refine_kernel<39, 32, 8, 1, 768, 1, false>", "gmm_refine_kernel<39, 32, 8, 2, 768, 1, false>", "gmm_refine_kernel<39, 8, 8, 1, 768, 1, false>",
              "gmm_refine_kernel<63, 32, 4, 1, 512, 1, false>", "gmm_refine_kernel<47, 32, 4, 2, 768, 1, false>", "gmm_refine_kernel<9, 32, 8, 1, 768, 1, false>", "gmm_refine_kernel<39, 32, 8, 2, 768, 1, true>", "gmm_drain_kernel<39, 32, 8, 2, 768, 1>",
              "gmm_prefilter16_kernel<3, 4>", "gmm_prefilter16_kernel<4, 4>", "decode_words_kernel<3, 3, false, false, 1024>", "decode_words_kernel<1, 3, false, false, 1024>",
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


HANDOFF_SRC = r'''
#include <hip/hip_runtime.h>
struct __attribute__((aligned(16))) E { unsigned a, b; double s; };
__global__ void handoff(E* ring, double* table, const unsigned* perm, double* out) {
  const unsigned lane = threadIdx.x, slot = perm[lane];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the permutation has arrived: nothing outstanding from here on)
  ring[slot] = E{lane, lane * 3u, (double)lane};
  table[lane] = 1.0;  // a younger store nobody reads back
  asm volatile("" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, SCOPE);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SCOPE);
  asm volatile("" ::: "memory");
  const E e = ring[lane];  // stored by another lane of this wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, SCOPE);
  (void)__builtin_amdgcn_global_atomic_fmin_f64((__attribute__((address_space(1))) double*)(table + (lane ^ 1u)), e.s);
  out[lane] = e.s + e.a;
}
'''


def _between_store_and_readback(insts):
    """The instructions from the entry store (the 16-byte one) up to the first load behind it."""
    i = next(k for k, x in enumerate(insts) if x.startswith("global_store_dwordx4"))
    j = next(k for k in range(i + 1, len(insts)) if insts[k].startswith("global_load"))
    return insts[i + 1:j]


def test_wavefront_scope_handoff_needs_no_wait_on_gfx950(tmp_path):
    src = tmp_path / "handoff.hip"
    src.write_text(HANDOFF_SRC)
    seen = {}
    for scope in ("wavefront", "workgroup", "agent"):
        co = tmp_path / f"{scope}.co"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "--no-gpu-bundle-output", f'-DSCOPE="{scope}"', "-c", str(src),
                        "-o", str(co)], check=True)
        insts = next(v for k, v in isa_info.disassembly(str(co)).items() if "handoff" in k)
        gap = _between_store_and_readback(insts)
        seen[scope] = [x for x in gap if x.startswith(("s_waitcnt", "buffer_wbl2", "buffer_inv"))]
        # the atomic minimum follows the read-back's use; between the fence in front of it and the atomic no write-back either
        k = next(i for i, x in enumerate(insts) if x.startswith("global_atomic_min_f64"))
        if scope != "agent":
            assert not any(x.startswith(("buffer_wbl2", "buffer_inv")) for x in insts[:k]), (scope, insts[:k])
    # one wave (and one CU: workgroup scope) needs nothing between the store and the other lane's load ...
    assert seen["wavefront"] == [] and seen["workgroup"] == [], seen
    # ... and the check would see it if it did: across CUs the compiler writes back, waits and invalidates
    assert any(x.startswith("buffer_wbl2") for x in seen["agent"]) and any("vmcnt(0)" in x for x in seen["agent"]), seen


def test_refinement_source_orders_its_lists_by_fences_not_by_counted_waits():
    """gmm_refine_kernel and gmm_drain_kernel: ONE explicit wait each (vmcnt(0) behind the LDS fill), every list read-back and the
    atomic minimum behind wavefront-scope fences.  A hand-counted vmcnt(N) in a batch loop (rounds 3-4) fails here."""
    import re
    text = open(os.path.join(ROOT, "speechrecognition_amd", "csrc", "gmm_prefilter.hip")).read()
    cut = [text.index("void gmm_refine_kernel("), text.index("void gmm_drain_kernel("), text.index("__global__ void transpose_feats_kernel")]
    for a, b, read_back in ((cut[0], cut[1], "const RingEntry en = ring[at];"), (cut[1], cut[2], "const RingEntry en = ring[j * kRingEntries")):
        code = "\n".join(line.split("//")[0] for line in text[a:b].splitlines())
        waits = re.findall(r"__builtin_amdgcn_s_waitcnt\(([^)]*)\)", code)
        assert waits == ["0x0F70"], waits
        assert "s_waitcnt" not in re.sub(r"__builtin_amdgcn_s_waitcnt\(0x0F70\)", "", code), "a wait in inline asm?"
        before = code[:code.index(read_back)]
        assert before.rindex('__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")') > before.rindex("RingEntry{"), "read-back must sit behind the fence pair"
    refine = "\n".join(line.split("//")[0] for line in text[cut[0]:cut[1]].splitlines())
    at = refine.rindex("global_atomic_fmin_f64")
    assert '__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront")' in refine[at - 400:at]
