import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_attn as t
cuda = torch.device("cuda:0")
print("spike N=1400", t._run(cuda, 1, 2, 1400, spike=True))
print("spike N=700", t._run(cuda, 1, 2, 700, spike=True))
print("overflow", t._run(cuda, 1, 2, 1122, spike=True, q_scale=6.0, relative=True))
