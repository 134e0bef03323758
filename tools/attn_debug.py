import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from landiff_amd import ops
torch.manual_seed(0)
B,H,N=1,1,int(os.environ.get("N","128"))
q = torch.randn(B,H,N,64).cuda().bfloat16(); k = torch.randn(B,H,N,64).cuda().bfloat16(); v = torch.randn(B,H,N,64).cuda().bfloat16()
Npad=(N+127)//128*128
def pack(x):
    o=torch.zeros(B,H,Npad,64,device="cuda",dtype=x.dtype); o[:,:,:N]=x; return o
out=torch.zeros(B,N,H*64,device="cuda",dtype=torch.bfloat16)
ops.attn_fwd(pack(q),pack(k),pack(v).transpose(2,3).contiguous(),out,N,N,0.125)
s=(q.float()@k.float().transpose(-1,-2))*0.125
ref=(torch.softmax(s,-1)@v.float()).permute(0,2,1,3).reshape(B,N,H*64)
err=(out.float()-ref).abs()
print("max err",err.max().item(),"mean",err.mean().item(), "ratio out/ref median", (out.float()/ref).median().item())
print(out[0,:4,:6].float()); print(ref[0,:4,:6])
