"""Builds speechrecognition_amd/csrc/build/variants/libsrgpu_wstamps.so: decode_words_kernel with s_memtime stamps around the phases of
its frame loop (diagnostic; the stamps overwrite traceback scores 1..32 of every utterance: words and back pointers stay right, the
traceback SCORES do not).  The product source is patched in place, compiled as a variant and restored; read the stamps with
tools/words_stamps_r4.py (SRGPU_LIB=<the variant>).  profiles/r4_words_stamps.txt."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, 'speechrecognition_amd/csrc/viterbi_words.hip')
shutil.copy(p, '/tmp/viterbi_words_clean.hip')
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
rep("#else\n  for (uint32_t t = 1; t <= T; t++) {\n",
    """#else
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#define SR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_sum[k] += now_ - stamp_last; stamp_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
  for (uint32_t t = 1; t <= T; t++) {
    SR_STAMP(0);
""")
rep("    // ---- A: the new hypotheses of the lane's words", "    SR_STAMP(1);\n    // ---- A: the new hypotheses of the lane's words")
rep("    // ---- B: block minima through LDS ds_min_f64 cells", "    SR_STAMP(2);\n    // ---- B: block minima through LDS ds_min_f64 cells")
rep("    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next row have landed; the barrier publishes them\n    __syncthreads();\n    // Row t + 2",
    "    SR_STAMP(3);\n    __builtin_amdgcn_s_waitcnt(0x0F70);\n    SR_STAMP(4);\n    __syncthreads();\n    SR_STAMP(5);\n    // Row t + 2")
rep("    if (t > 1) flush_pending(t - 1);\n    m_we = we_alive", "    SR_STAMP(6);\n    if (t > 1) flush_pending(t - 1);\n    m_we = we_alive")
rep("      r_prev = r; r = r_next;\n    }\n  }\n#endif\n  __syncthreads();\n  if (T > 0) flush_pending(T);\n",
    """      r_prev = r; r = r_next;
    }
    SR_STAMP(7);
  }
#endif
  __syncthreads();
  if (T > 0) flush_pending(T);
  __syncthreads();
  if (lane == 0 && wave < 4) for (int k = 0; k < 8 && 8 * wave + k + 1 <= T; k++) a.tb_score[tb0 + 1 + 8 * wave + k] = (double)stamp_sum[k];
""")
open(p, 'w').write(s)
try:
    subprocess.check_call([sys.executable, 'tools/build_variant.py', 'wstamps'], cwd=ROOT)
finally:
    shutil.copy('/tmp/viterbi_words_clean.hip', p)
