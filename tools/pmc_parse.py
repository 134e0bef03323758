"""Sums rocprofv3 --pmc counter_collection.csv per kernel name / counter (last launch of each kernel)."""
import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "ld_" not in k: continue
            print(d, k)
            for c, v in sorted(cs.items()):
                print(f"   {c:32s} n={len(v)} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}")
