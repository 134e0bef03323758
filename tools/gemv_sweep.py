"""Sweep rows-per-batch (LD_GEMV_R) and the workgroup cap (LD_GEMV_WGS) of the register-resident decode GEMV on the four LLM shapes
(tools/gemv_shapes.py in child processes: the knobs are read once).  python tools/gemv_sweep.py [shapes...]"""
import os, re, subprocess, sys
shapes = sys.argv[1:] or ["qkv", "wo", "gated", "w2"]
RS = {"qkv": [1, 2, 4, 8], "wo": [1, 2, 4, 8], "gated": [1, 2, 4], "w2": [1, 2]}
CAPS = [512, 768, 1024, 1536, 2048, 4096]
for sh in shapes:
    print(f"# {sh}: us per launch (rows = R, columns = workgroup cap)")
    print("R    " + "".join(f"{c:8d}" for c in CAPS))
    for R in RS[sh]:
        row = []
        for cap in CAPS:
            e = dict(os.environ, SHAPE=sh, LD_GEMV_R=str(R), LD_GEMV_WGS=str(cap))
            r = subprocess.run([sys.executable, "tools/gemv_shapes.py"], env=e, capture_output=True, text=True)
            m = re.search(r":\s+([\d.]+) us per layer-op", r.stdout)
            row.append(float(m.group(1)) if m else float("nan"))
        print(f"{R:<4d} " + "".join(f"{v:8.2f}" for v in row), flush=True)
