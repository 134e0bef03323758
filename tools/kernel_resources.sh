#!/bin/bash
# Register / occupancy table of one kernel file: tools/kernel_resources.sh ld_llm.hip [filter] [extra hipcc flags]
f=$1; filt=${2:-.}; shift; shift
cd "$(dirname "$0")/../landiff_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ra_$$.o 2>&1 \
  | python3 -c '
import re, sys, subprocess
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k.split()[0]] = v
for r in rows:
    try: name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception: name = r["name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*", "", name)
    print("%-70s vgpr %4s agpr %3s sgpr %3s scratch %4s occ %s lds %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
' | grep -E "$filt"
rm -f /tmp/ra_$$.o
