"""Phase timeline of the persistent decode launch (variant built with -DLD_FUSED_TRACE=<workgroup>): 100 MHz wall clock of one
workgroup at the phase boundaries of layer 1, averaged over steps.
  tools/build_variant.sh trace ld_llm_fused.hip -DLD_FUSED_TRACE=0
  LANDIFF_HIP_LIB=landiff_amd/variants/lib_trace.so python tools/llm_fused_trace.py"""
import sys
import torch
sys.path.insert(0, ".")
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec

dev = torch.device("cuda:0")
cfg = LLMConfig()
sd = init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev)
run = LLMRunner(sd, cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
run.fused_ctl = torch.zeros(512 + 2 * 12 * 512, device=dev, dtype=torch.int32)
run.sample(text, guidance_scale=7.5, seed=42, num_frames=2, mode="fused")
LABELS = ["qkv: load x + norm", "qkv: batches", "qkv: drain", "barrier 1 (+ wo request)", "attention", "attn: drain", "barrier 2",
          "combine", "comb: drain", "barrier 3", "wo: load x", "wo: batches", "wo: drain", "barrier 4", "w13: load x + norm",
          "w13: batches", "w13: drain", "barrier 5", "w2: load x", "w2: batches", "w2: drain", "barrier 6"]
acc = torch.zeros(len(LABELS), dtype=torch.float64)
n = 0
for it in range(200):
    run._decode_forward()
    torch.cuda.synchronize()
    ts = run.fused_ctl[288:288 + 2 * 23].cpu().view(torch.int64).double()
    acc += ts[1:23] - ts[0:22]
    n += 1
acc /= n
tot = acc.sum().item()
print(f"layer 1 of a step, workgroup trace, averaged over {n} steps (pos {int(run.pos)}): total {tot / 100:.2f} us")
for lab, v in zip(LABELS, acc.tolist()):
    print(f"  {lab:22s} {v / 100:7.2f} us")

# all workgroups: when did each arrive at / leave the six barriers of layer 1 (last step)
tb = run.fused_ctl[512:512 + 2 * 12 * 256].cpu().view(torch.int64).double().reshape(256, 12) / 100.0      # us
t0 = tb[:, 0].min()
print("barrier: arrival spread over workgroups (first .. last, us from the first arrival at barrier 1), exit after the last arrival")
for k in range(6):
    arr, ex = tb[:, 2 * k] - t0, tb[:, 2 * k + 1] - t0
    print(f"  barrier {k + 1}: arrivals {arr.min():7.2f} .. {arr.max():7.2f} (median {arr.median():7.2f}), exits {ex.min():7.2f} .. {ex.max():7.2f};"
          f"  last arrival -> median exit {ex.median() - arr.max():5.2f} us")

# who are the stragglers?  phase time of every workgroup = arrival at barrier k+1 - exit of barrier k, by XCD (workgroup id & 7)
names = ["attention", "combine", "wo", "w13", "w2"]
for k, nm in enumerate(names):
    dur = tb[:, 2 * (k + 1)] - tb[:, 2 * k + 1]
    by_xcd = [dur[x::8] for x in range(8)]
    print(f"  {nm:9s} per-XCD mean: " + " ".join(f"{d.mean():6.2f}" for d in by_xcd) + f"   | all: min {dur.min():.2f} median {dur.median():.2f} max {dur.max():.2f}")
    if nm in ("w13", "w2"):
        order = torch.argsort(dur, descending=True)[:12]
        print("            slowest workgroups: " + " ".join(f"{int(i)}({dur[i]:.1f})" for i in order))
