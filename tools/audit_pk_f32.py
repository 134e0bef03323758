"""ISA audit of a built library (default: the shipped landiff_amd/liblandiff_hip.so): no packed-fp32 VALU instruction may let its LOW
lane read the HIGH register of an operand pair (`op_sel:[..1..]` on v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).

Why: round 5 found that exactly this operand form returns a wrong value -- the swizzled operand reads as 0.0 -- in lanes 48-63 of the
wave when MFMA-issuing waves of another kernel share the SIMD (tools/probe/pk_f32_coresidency.hip, profiles/
r05_pk_f32_coresidency_probe.txt: 0 mismatches of 6.3e9 on a quiet GPU, 2e4 - 2e6 beside the attention kernels, every one of them in
lanes 48-63; the same sequence without the swizzle: 0).  hipcc's SLP vectoriser emits the form for expressions like RoPE's
`a*c - b*s` / `a*s + b*c`; the kernel files are built with -fno-slp-vectorize where it did, and this audit (a CPU test:
tests/test_cabi_and_host.py) keeps a toolchain bump or a new kernel from bringing it back unnoticed.

Works on the built .so: the device code objects are taken out of its .hip_fatbin section (one clang offload bundle per translation
unit), disassembled with llvm-objdump, and scanned.  usage: python tools/audit_pk_f32.py [lib.so]   -> exit code 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
RISKY = re.compile(r"v_pk_(?:add|mul|fma)_f32\b.*\bop_sel:\[[01,]*1")


def device_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    # (-O binary with an explicit output: `objcopy --dump-section=... lib` without one REWRITES lib in place -- a process that has the
    #  library mapped, e.g. the test run that called us, then dies of SIGBUS on its next page fault into it)
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        chunk = os.path.join(tmp, f"bundle{i}.bin")
        open(chunk, "wb").write(blob[a:b])
        co = os.path.join(tmp, f"dev{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={chunk}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def audit(lib):
    """-> (number of code objects, number of kernels, [(kernel, instruction text)])"""
    findings, nk = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        objs = device_objects(lib, tmp)
        for co in objs:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    kernel = m.group(1); nk += 1
                    continue
                if RISKY.search(line):
                    findings.append((kernel, line.split("//")[0].strip()))
    return len(objs), nk, findings


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "landiff_amd", "liblandiff_hip.so")
    n_obj, n_k, found = audit(lib)
    print(f"{lib}: {n_obj} device code objects, {n_k} functions, {len(found)} packed-fp32 instructions whose low lane reads a high register")
    for k, ins in found[:40]:
        print(f"  {k}: {ins}")
    sys.exit(1 if found or n_obj == 0 else 0)
