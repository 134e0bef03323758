"""ISA audit of ld_attn_q128.hip (no GPU needed: hipcc cross-compiles).

The kernel keeps O^T, the Q^T fragments and the K / V^T fragments in accumulator registers a[0:227] that only its asm statements
name; the compiler does not know they are live.  That is safe only while the compiler itself never touches a[] in this kernel
and never spills (cdna_hip_programming.md 5.7 item 4).  This script rebuilds the file with -save-temps and checks, for every
q128 kernel in it:
  * no scratch, no VGPR / SGPR spills, 228 AGPRs and <= 256 architectural VGPRs in the kernel descriptor,
  * no v_accvgpr_* instruction outside the ;;#ASMSTART / ;;#ASMEND blocks (i.e. none emitted by the compiler),
  * the hot loop (eight halves per trip) holds exactly 576 MFMAs, 512 v_exp_f32, 256 v_cvt_pk_bf16_f32, 64 ds_read_b128 into
    a[], 16 LDS-DMA loads, and the VALU work is interleaved with the MFMAs: never more than MAX_RUN VALU instructions between
    two MFMAs (hipcc once sank all 32 packs of a half in front of the first PV MFMA).
Exit status 0 and a one-line summary per kernel on success; raises AssertionError otherwise.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "landiff_amd", "csrc", "ld_attn_q128.hip")
MAX_RUN = 4


def build(tmp):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize",
           "-save-temps=obj", "-c", SRC, "-o", os.path.join(tmp, "q128.o")]
    subprocess.run(cmd, check=True, cwd=os.path.dirname(SRC), capture_output=True)
    s = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
    assert len(s) == 1, s
    return open(os.path.join(tmp, s[0])).read()


def kernels(asm):
    """name -> (body lines, metadata dict)"""
    out = {}
    for m in re.finditer(r"^(_ZN\S*ld_attn_q128\S*kernel\S*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    meta = {}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", asm, re.S):
        meta[m.group(2)] = dict(agpr=int(m.group(1)), scratch=int(m.group(3)), sgpr_spill=int(m.group(4)), vgpr=int(m.group(5)),
                                vgpr_spill=int(m.group(6)))
    return out, meta


def audit(asm):
    bodies, meta = kernels(asm)
    assert len(bodies) >= 1 and set(bodies) <= set(meta), (list(bodies), list(meta))
    report = []
    for name, lines in bodies.items():
        md = meta[name]
        assert md["scratch"] == 0 and md["vgpr_spill"] == 0 and md["sgpr_spill"] == 0, (name, md)
        arch = md["vgpr"] - md["agpr"]           # .vgpr_count is the unified total on gfx950 (accum_offset = architectural part)
        assert md["agpr"] == 228 and md["vgpr"] <= 512 and 0 < arch <= 256, (name, md)
        in_asm, stray = False, []
        for ln in lines:
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif not in_asm and ("v_accvgpr" in ln or re.search(r"\ba\[?\d", ln.split(";")[0])):
                stray.append(ln.strip())
        assert not stray, (name, "compiler-emitted accumulator-register traffic", stray[:5])
        assert not any("scratch_" in ln for ln in lines), name
        # hot loop: the first inner loop of the kernel (the unmasked trips)
        start = next(i for i, ln in enumerate(lines) if "Inner Loop Header" in ln)
        end = next(i for i in range(start + 1, len(lines)) if re.search(r"s_cbranch_scc0\s+\.LBB\d+_\d+", lines[i]))
        loop = [ln.split(";")[0].strip() for ln in lines[start:end] if ln.strip() and not ln.strip().startswith(";")]
        cnt = lambda pat: sum(1 for ln in loop if re.match(pat, ln))
        counts = dict(mfma=cnt(r"v_mfma_f32_16x16x32_bf16"), exp=cnt(r"v_exp_f32"), pack=cnt(r"v_cvt_pk_bf16_f32"),
                      ds=cnt(r"ds_read_b128 a\["), dma=cnt(r"buffer_load_dwordx4 .* lds"))
        assert counts == dict(mfma=576, exp=512, pack=256, ds=64, dma=16), (name, counts)
        # every vector instruction of the loop must come from an asm statement: a compiler-emitted VALU write next to an asm MFMA
        # gets no wait states (the all-ones fragment was once re-materialised by v_mov_b64 right in front of its MFMA)
        in_asm, loose = False, []
        for ln in lines[start:end]:
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif not in_asm and re.match(r"\s*(v_|ds_)", ln):
                loose.append(ln.strip())
        assert not loose, (name, "compiler-emitted vector instructions in the hot loop", loose[:5])
        run, worst = 0, 0
        for ln in loop:
            if ln.startswith("v_mfma"):
                run = 0
            elif ln.startswith("v_"):
                run += 1
                worst = max(worst, run)
        assert worst <= MAX_RUN, (name, f"{worst} VALU instructions between two MFMAs")
        report.append(f"{name}: vgpr {arch}, agpr {md['agpr']}, scratch 0, no compiler a[] traffic; loop {counts}, "
                      f"longest VALU run between MFMAs {worst}")
    return report


def main():
    with tempfile.TemporaryDirectory() as tmp:
        for line in audit(build(tmp)):
            print(line)


if __name__ == "__main__":
    sys.exit(main())
