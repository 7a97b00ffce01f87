"""Builds speechrecognition_amd/csrc/build/variants/libsrgpu_bgstamps.so: bigram_kernel (register layout, the BASELINE configurations) with
s_memtime stamps around its steps (diagnostic; the stamps replace the first output scores of every utterance, the words are garbage).  Read
them with tools/bigram_stamps_r4.py (SRGPU_LIB=<the variant>).  profiles/r4_bigram_steps.txt."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, 'speechrecognition_amd/csrc/viterbi_bigram.hip')
shutil.copy(p, '/tmp/viterbi_bigram_clean.hip')
s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
rep("  for (uint64_t t = 1; t <= T; t++) {\n    // ---- 1 bigramRecombination", """  unsigned long long stamp_sum[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, stamp_last = __builtin_amdgcn_s_memtime();
#define SR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_sum[k] += now_ - stamp_last; stamp_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
  for (uint64_t t = 1; t <= T; t++) {
    // ---- 1 bigramRecombination""")
rep("    const float U = wg_min(lu, red_tmp);\n", "    const float U = wg_min(lu, red_tmp);\n    SR_STAMP(0);\n")
rep("    uint32_t ne_all;\n    const uint32_t keep_pos = wg_excl_scan(n_keep, scan_tmp, &ne_all);\n", "    uint32_t ne_all;\n    const uint32_t keep_pos = wg_excl_scan(n_keep, scan_tmp, &ne_all);\n    SR_STAMP(1);\n")
rep("    const float best_start = wg_min(lmin, red_tmp);\n", "    const float best_start = wg_min(lmin, red_tmp);\n    SR_STAMP(2);\n")
rep("    // ---- 3 expandHypotheses + addAcousticScores, one thread per POSITION", "    SR_STAMP(3);\n    // ---- 3 expandHypotheses + addAcousticScores, one thread per POSITION")
rep("    const float best_score = wg_min(lbest, red_tmp);\n", "    SR_STAMP(4);\n    const float best_score = wg_min(lbest, red_tmp);\n    SR_STAMP(5);\n")
rep("    const uint32_t cl = (n_L + kBgThreads - 1) / kBgThreads;\n    const uint32_t i_lo = tid * cl", "    SR_STAMP(6);\n    const uint32_t cl = (n_L + kBgThreads - 1) / kBgThreads;\n    const uint32_t i_lo = tid * cl")
rep("    // ---- 5 mergeSilenceToBigramNodes.", "    SR_STAMP(7);\n    // ---- 5 mergeSilenceToBigramNodes.")
rep("      n_hist = tot_ends - *n_pairs;", "      SR_STAMP(8);\n      n_hist = tot_ends - *n_pairs;")
rep("    n_we = n_hist;\n    __syncthreads();\n", "    n_we = n_hist;\n    SR_STAMP(9);\n    __syncthreads();\n    SR_STAMP(10);\n")
rep("  // ---- traceback (:420-436): first minimum in list order", "  if (tid == 0 && T >= 14) for (int k = 0; k < 11; k++) a.out_score[f0 + u + 1 + k] = (float)((double)stamp_sum[k] / (double)T);\n  if (T >= 14) { if (tid == 0) { a.out_count[u] = 12; a.out_flags[u] = 0; } return; }\n  // ---- traceback (:420-436): first minimum in list order")
open(p, 'w').write(s)
try:
    subprocess.check_call([sys.executable, 'tools/build_variant.py', 'bgstamps'], cwd=ROOT)
finally:
    shutil.copy('/tmp/viterbi_bigram_clean.hip', p)
