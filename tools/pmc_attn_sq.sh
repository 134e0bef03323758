#!/bin/bash
# SQ-side counters of the DiT attention launch for the two wave tiles (q64: two waves per SIMD, the default; q128: one wave per
# SIMD, LD_ATTN_Q128=2): two rocprofv3 --pmc passes per variant, the program itself after `--`.  Summary to $1.
#   matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves per SIMD)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=${1:-gpurun_out/r04_attn_pmc_sq.txt}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rm -rf /tmp/pmcsq_*
rocprofv3 --pmc $P1 -d /tmp/pmcsq_q64_1 -o a --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
rocprofv3 --pmc $P2 -d /tmp/pmcsq_q64_2 -o a --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
export LD_ATTN_Q128=2
rocprofv3 --pmc $P1 -d /tmp/pmcsq_q128_1 -o a --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
rocprofv3 --pmc $P2 -d /tmp/pmcsq_q128_2 -o a --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
with open(sys.argv[1], "w") as o:
    o.write("# rocprofv3 --pmc (two passes per variant) -- python3 tools/attn_one.py : DiT attention launch B=2,H=30,N=17776,D=64; q128 = LD_ATTN_Q128=2\n")
    for tag, wps in (("q64", 2), ("q128", 1)):
        vals = {}
        for d in sorted(glob.glob(f"/tmp/pmcsq_{tag}_*")):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "attn" in r["Kernel_Name"]:
                        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                        kern = r["Kernel_Name"].split("::")[-1].split("(")[0]
        if not vals:
            o.write(f"{tag}: no counters collected\n"); continue
        o.write(f"{kern} ({wps} wave(s) per SIMD)\n")
        for k in sorted(vals):
            o.write(f"   {k:28s} n={len(vals[k])} last={vals[k][-1]:.4e}\n")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_WAVE_CYCLES" in vals:
            o.write(f"   matrix pipe busy = {vals['SQ_VALU_MFMA_BUSY_CYCLES'][-1] / (4 * vals['SQ_WAVE_CYCLES'][-1] / wps):.3f} of the resident cycles\n")
print(open(sys.argv[1]).read())
PY
