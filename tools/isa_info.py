"""Properties of the gfx950 code objects inside the built objects, read without a GPU (llvm-objdump / llvm-readelf from
/opt/rocm/lib/llvm/bin): per kernel the VGPR / AGPR / SGPR counts, spills, scratch (private segment) and LDS sizes from the
AMDGPU metadata note, and the disassembly for checks on instruction order (tests/test_isa_cpu.py).

usage: python3 tools/isa_info.py [object-name ...]      e.g.  python3 tools/isa_info.py gmm_prefilter viterbi_words
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "speechrecognition_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(name: str, workdir: str) -> str:
    """Unbundles the gfx950 ELF out of speechrecognition_amd/csrc/build/<name>.hip.o into workdir; returns its path."""
    src = os.path.join(OBJDIR, name + ".hip.o")
    local = os.path.join(workdir, name + ".o")
    shutil.copy(src, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(workdir):
        if f.startswith(name + ".o.") and f.endswith("gfx950"):
            return os.path.join(workdir, f)
    raise RuntimeError(f"no gfx950 code object in {src}")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True, check=True).stdout
    return [re.sub(r"^void ", "", d).replace("srgpu::", "").split("(")[0] for d in out.splitlines()]


def kernel_metadata(co: str) -> dict:
    """{demangled kernel name (template arguments kept, parameters dropped): {field: value}} from the NT_AMDGPU_METADATA note."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True, capture_output=True, check=True).stdout
    kernels, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key in ("agpr_count", "args") and line.lstrip().startswith("- "):
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                       "group_segment_fixed_size", "max_flat_workgroup_size", "name", "symbol", "uses_dynamic_stack"):
            cur[key] = int(val) if re.fullmatch(r"-?\d+", val) else val
    names = demangle([k.get("name", "?") for k in kernels])
    return {n: k for n, k in zip(names, kernels)}


def disassembly(co: str) -> dict:
    """{demangled kernel name: [instruction lines]}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True, capture_output=True, check=True).stdout
    raw, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = raw.setdefault(m.group(1), [])
        elif cur is not None and line.startswith("\t"):
            cur.append(line.strip().split("//")[0].strip())
    names = demangle(list(raw))
    return {n: v for n, v in zip(names, raw.values())}


def main():
    names = sys.argv[1:] or sorted(f[:-6] for f in os.listdir(OBJDIR) if f.endswith(".hip.o"))
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            md = kernel_metadata(code_object(name, tmp))
            for k, v in sorted(md.items()):
                print(f"{k[:84]:84s} vgpr {v.get('vgpr_count', 0):3d} agpr {v.get('agpr_count', 0):3d} sgpr {v.get('sgpr_count', 0):3d} "
                      f"spill v{v.get('vgpr_spill_count', 0)} s{v.get('sgpr_spill_count', 0)} scratch {v.get('private_segment_fixed_size', 0)} "
                      f"lds {v.get('group_segment_fixed_size', 0)}")


if __name__ == "__main__":
    main()
