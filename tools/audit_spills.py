"""ISA audit of a built library (default: the shipped landiff_amd/liblandiff_hip.so): no register spill traffic (scratch_load /
scratch_store) inside a loop that issues MFMAs.

Why: the pipelined kernels keep LDS-DMA and global loads in flight across iterations and wait for them with COUNTED s_waitcnt vmcnt(N).
A spilled register's reload is a vector-memory load too, and hipcc waits for it with s_waitcnt vmcnt(0): one reload inside a K loop
drains the whole prefetch queue every iteration.  Round 6 found exactly that in three instantiations of the MXFP8 GEMM (one per-lane
DMA offset spilled; dense 200 -> 166 us, 4h->h 672 -> 497 us once it was gone; profiles/r06_mx_serial_step_kernel_stats*.csv), and
the register files of the attention and GEMM kernels are full by design, so a toolchain bump or an innocent edit can bring one back.

A loop = the address range of a backward branch (s_cbranch_* / s_branch to a lower address); flagged = a scratch_* instruction whose
innermost enclosing MFMA loop has no MFMA loop nested inside it (a reload in an outer tile / block walk runs once per tile: fine).  Works on the built .so like tools/audit_pk_f32.py.
usage: python tools/audit_spills.py [lib.so]   -> exit code 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audit_pk_f32 import LLVM, device_objects  # noqa: E402

INS = re.compile(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):")
BR = re.compile(r"^\s+(s_cbranch_\w+|s_branch)\s+(\d+)\s.*//\s*([0-9A-Fa-f]+):.*<[^>]*\+0x([0-9A-Fa-f]+)>")


def audit(lib):
    """-> (code objects, functions, [(function, loop start, loop end, n scratch ops, first one's text)])"""
    findings, nk = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        objs = device_objects(lib, tmp)
        for co in objs:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
            funcs, cur = [], None
            for line in dis.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
                if m:
                    cur = {"name": m.group(2), "base": int(m.group(1), 16), "ins": []}
                    funcs.append(cur); nk += 1
                    continue
                if cur is None:
                    continue
                m = INS.match(line)
                if not m:
                    continue
                addr = int(m.group(2), 16)
                tgt = None
                b = BR.match(line)
                if b:
                    tgt = cur["base"] + int(b.group(4), 16)
                cur["ins"].append((addr, m.group(1), tgt, line.split("//")[0].strip()))
            for f in funcs:
                ins = f["ins"]
                mf = [a for a, op, _, _ in ins if op.startswith("v_mfma") or op.startswith("v_smfmac")]
                sc = [(a, t) for a, op, _, t in ins if op.startswith("scratch_")]
                if not mf or not sc:
                    continue
                loops = sorted({(t, a) for a, _, t, _ in ins if t is not None and t <= a})
                mf_loops = [(lo, hi) for lo, hi in loops if any(lo <= a <= hi for a in mf)]
                for a, text in sc:
                    around = [(hi - lo, lo, hi) for lo, hi in mf_loops if lo <= a <= hi]
                    if not around:
                        continue
                    _, lo, hi = min(around)              # the innermost MFMA loop around this spill instruction
                    # an OUTER loop (persistent tile / block walk) whose MFMAs sit in nested loops pays a reload once per tile: fine
                    nested = [(l2, h2) for l2, h2 in mf_loops if lo <= l2 and h2 <= hi and (l2, h2) != (lo, hi)]
                    direct = [m for m in mf if lo <= m <= hi and not any(l2 <= m <= h2 for l2, h2 in nested)]
                    if direct and not nested:
                        n_here = sum(1 for a2, _ in sc if lo <= a2 <= hi)
                        findings.append((f["name"], lo - f["base"], hi - f["base"], n_here, text))
    best = {}                                   # one line per (function, loop)
    for name, lo, hi, n, text in findings:
        best.setdefault((name, lo, hi), (name, lo, hi, n, text))
    return len(objs), nk, sorted(best.values())


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "landiff_amd", "liblandiff_hip.so")
    n_obj, n_k, found = audit(lib)
    print(f"{lib}: {n_obj} device code objects, {n_k} functions, {len(found)} MFMA loops with spill traffic")
    for name, lo, hi, n, text in found[:40]:
        print(f"  {name}: loop +0x{lo:x} .. +0x{hi:x} ({hi - lo} bytes): {n} scratch instruction(s), e.g. {text}")
    sys.exit(1 if found or n_obj == 0 else 0)
