"""Builds speechrecognition_amd/csrc/build/variants/libsrgpu_bgstamps.so: bigram_kernel with s_memtime stamps around its steps (diagnostic;
the stamps replace the traceback scores, the words are garbage).  Read them with tools/bigram_stamps_r3.py (SRGPU_LIB=<the variant>).
profiles/r3_bigram_steps.txt."""
import shutil, subprocess, sys
p='/root/repo/speechrecognition_amd/csrc/viterbi_bigram.hip'
shutil.copy(p,'/tmp/viterbi_bigram_clean.hip')
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a[:50])
    s=s.replace(a,b)
rep("  for (uint64_t t = 1; t <= T; t++) {\n    // ---- 1 bigramRecombination","""  unsigned long long stamp_sum[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, stamp_last = __builtin_amdgcn_s_memtime();
#define SR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stamp_sum[k] += now_ - stamp_last; stamp_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
  for (uint64_t t = 1; t <= T; t++) {
    // ---- 1 bigramRecombination""")
rep("    const float U = wg_min(lu, red_tmp);\n","    const float U = wg_min(lu, red_tmp);\n    SR_STAMP(0);\n")
rep("    const float best_start = wg_min(lmin, red_tmp);\n","    const float best_start = wg_min(lmin, red_tmp);\n    SR_STAMP(1);\n")
rep("    // ---- 3 expandHypotheses + addAcousticScores, one thread per POSITION","    SR_STAMP(2);\n    // ---- 3 expandHypotheses + addAcousticScores, one thread per POSITION")
rep("    __syncthreads();  // every old state has been read\n","    SR_STAMP(3);\n    __syncthreads();  // every old state has been read\n    SR_STAMP(4);\n")
rep("    // ---- 4 pruneStatesAndFindWordEnds: the acoustic beam per position","    SR_STAMP(5);\n    // ---- 4 pruneStatesAndFindWordEnds: the acoustic beam per position")
rep("    const uint32_t cl = (n_L + kBgThreads - 1) / kBgThreads;\n    const uint32_t i_lo = tid * cl","    SR_STAMP(6);\n    const uint32_t cl = (n_L + kBgThreads - 1) / kBgThreads;\n    const uint32_t i_lo = tid * cl")
rep("    // ---- 5 mergeSilenceToBigramNodes.","    SR_STAMP(7);\n    // ---- 5 mergeSilenceToBigramNodes.")
rep("    // ---- 6 addBookKeepingEntries","    SR_STAMP(8);\n    // ---- 6 addBookKeepingEntries")
rep("    n_we = n_hist;\n    __syncthreads();\n","    n_we = n_hist;\n    __syncthreads();\n    SR_STAMP(9);\n")
# dump: thread 0 of each workgroup writes into out_score region? use out_score[f0+u + k] floats (first 10 entries) -- traceback output is garbage then
rep("  // ---- traceback (:420-436): first minimum in list order","  if (tid == 0 && T >= 12) for (int k = 0; k < 10; k++) a.out_score[f0 + u + 1 + k] = (float)((double)stamp_sum[k] / (double)T);\n  if (T >= 12) { if (tid == 0) { a.out_count[u] = 11; a.out_flags[u] = 0; } return; }\n  // ---- traceback (:420-436): first minimum in list order")
open(p,'w').write(s)
try:
    subprocess.check_call([sys.executable,'tools/build_variant.py','bgstamps'],cwd='/root/repo')
finally:
    shutil.copy('/tmp/viterbi_bigram_clean.hip',p)
