import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev="cuda"; M=35552
torch.manual_seed(0)
for K in (1920, 7680):
    x=(torch.randn(M,K,device=dev)).to(torch.bfloat16)
    for N in (1536, 1792, 1920, 2048):
        w=(torch.randn(N,K,device=dev)*0.02).to(torch.bfloat16); b=torch.randn(N,device=dev).to(torch.bfloat16)
        o=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
        f=lambda: ops.gemm(x,w,out=o,bias=b)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(f"K={K} N={N}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s  ({ms*1e3/(N/256):.1f} us per tile column)")
