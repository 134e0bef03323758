#!/bin/bash
# HBM-side traffic of the DiT attention launch: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: the TCC slots do not
# fit both), the program itself after `--` (MI355X_MICROARCH.md: HBM / rocprofv3 PMC slots).  Writes profiles-style summary to $1.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=${1:-gpurun_out/r04_attn_pmc_hbm.txt}
rm -rf /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch -o f --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_write -o w --output-format csv -- python3 tools/attn_one.py > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
res = {}
for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
    vals = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr and "attn" in r["Kernel_Name"]:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    res[ctr] = vals
with open(sys.argv[1], "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/attn_one.py : DiT attention launch B=2,H=30,N=17776,D=64 (KiB per launch)\n")
    for ctr, vals in res.items():
        for k, v in vals.items():
            o.write(f"{k}\n   {ctr:12s} n={len(v)} last={v[-1]:.4e} mean={sum(v)/len(v):.4e}\n")
    k = list(res["FETCH_SIZE"])[0]
    fetch, write = res["FETCH_SIZE"][k][-1], list(res["WRITE_SIZE"].values())[0][-1]
    o.write(f"# traffic = (2 x FETCH_SIZE [gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM] + WRITE_SIZE) x 1024 = {(2 * fetch + write) * 1024:.4e} B per launch\n")
    o.write(f"TRAFFIC_BYTES {int((2 * fetch + write) * 1024)} KERNEL {k.split('::')[-1].split('(')[0]}\n")
print(open(sys.argv[1]).read())
PY
