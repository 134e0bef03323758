import torch
M, D = 35552, 1920
for (N, K) in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.addmm(bias, a, w.t(), out=out)
torch.cuda.synchronize()
