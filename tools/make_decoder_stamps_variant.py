"""Builds speechrecognition_amd/csrc/build/variants/libsrgpu_stamps.so: decode_fast_kernel with s_memtime stamps around the phases of its frame
loop (diagnostic; the stamps overwrite traceback scores 1..48 of every utterance).  The product source is patched in place, compiled as a
variant and restored; read the stamps with tools/decode_stamps_r3.py (SRGPU_LIB=<the variant>).  profiles/r3_decoder_diet.txt."""
import shutil, subprocess, sys
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p=os.path.join(ROOT,'speechrecognition_amd/csrc/viterbi_fast.hip')
shutil.copy(p,'/tmp/viterbi_fast_clean.hip')
s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a[:50]
    s=s.replace(a,b)
rep("  for (uint32_t t = 1; t <= T; t++) {\n    const uint32_t par = t & 1;","""#ifdef SR_DEC_STAMPS
  unsigned long long stamp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#define SR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stamp_sum[k] += now_ - stamp_last; stamp_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SR_STAMP(k) do {} while (0)
#endif
  for (uint32_t t = 1; t <= T; t++) {
    const uint32_t par = t & 1;""")
rep("    if (t > 1) flush_pending(t - 1);\n","    SR_STAMP(0);\n    if (t > 1) flush_pending(t - 1);\n")
rep("    // ---- A: candidates of every slot ---","    SR_STAMP(1);\n    // ---- A: candidates of every slot ---")
# mixed path: after loads of group, after compute of group
rep("          in[i - h].c0 = cell(slot_of(i) * 16u);\n        }\n","          in[i - h].c0 = cell(slot_of(i) * 16u);\n        }\n        if (h == 0) SR_STAMP(2); else SR_STAMP(4);\n")
rep("            default: break;  // kPad\n          }\n        }\n      }\n","            default: break;  // kPad\n          }\n        }\n        if (h == 0) SR_STAMP(3); else SR_STAMP(5);\n      }\n")
rep("    // ---- B: block minima through LDS ds_min_f64 cells","    SR_STAMP(6);\n    // ---- B: block minima through LDS ds_min_f64 cells")
rep("      publish_min_f64_lds(&c_best[par], my_best);\n    }\n    __syncthreads();\n","      publish_min_f64_lds(&c_best[par], my_best);\n    }\n    SR_STAMP(7);\n    __syncthreads();\n    SR_STAMP(8);\n")
rep("    const double limit = best + thr;\n    const bool we_alive","    SR_STAMP(9);\n    const double limit = best + thr;\n    const bool we_alive")
rep("    if (ROWS) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next row have landed; the barrier publishes them\n    __syncthreads();\n  }\n","    SR_STAMP(10);\n    if (ROWS) __builtin_amdgcn_s_waitcnt(0x0F70);\n    __syncthreads();\n    SR_STAMP(11);\n  }\n#ifdef SR_DEC_STAMPS\n  __syncthreads();\n  if ((tid & 63) == 0 && (wave == 0 || wave == 7 || wave == 10 || wave == 15)) { const int wi = wave == 0 ? 0 : wave == 7 ? 1 : wave == 10 ? 2 : 3; for (int k = 0; k < 12 && 12 * wi + k + 1 <= (int)T; k++) a.tb_score[tb0 + 1 + 12 * wi + k] = (double)stamp_sum[k]; }\n#endif\n")
open(p,'w').write(s)
try:
    subprocess.check_call([sys.executable,'tools/build_variant.py','stamps','-DSR_DEC_STAMPS'],cwd=ROOT)
finally:
    shutil.copy('/tmp/viterbi_fast_clean.hip',p)
