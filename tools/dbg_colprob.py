import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from spatten_amd import ops
torch.manual_seed(0)
B,H,d,P,ql = int(sys.argv[1]),int(sys.argv[2]),128,int(sys.argv[3]),int(sys.argv[4])
N=P+ql
dt=torch.bfloat16
K=torch.randn(B,H,N,d,device="cuda").to(dt); V=torch.randn(B,H,N,d,device="cuda").to(dt); Q=torch.randn(B,H,ql,d,device="cuda").to(dt)
cos,sin=ops.rope_table(N,d,dt,"cuda")
Kr=ops.rope_single(K,cos,sin)
lse=torch.empty(B,H,ql,2,dtype=torch.float32,device="cuda")
ops.attn_prefill(Q,Kr,V,N,cos,sin,P,causal=True,lse=lse)
acc=torch.zeros(H,N,dtype=torch.float32,device="cuda")
ops.importance_accumulate_prefill(acc,Q,Kr,N,cos,sin,P,lse,causal=True)
torch.cuda.synchronize()
Qr=ops.rope_single(Q,cos,sin,pos0=P)
want=0
for b in range(B):
    sc=(Qr[b,0].float()@Kr[b,0].float().T)/d**0.5
    sc=sc.masked_fill(torch.ones(ql,N,dtype=torch.bool,device="cuda").triu(P+1),float("-inf"))
    want=want+torch.softmax(sc,-1).sum(0)
a=acc[0].cpu().numpy(); w=want.cpu().numpy()
for k0 in range(0,N,64):
    print(k0, round(float(a[k0:k0+32].sum()),3), round(float(w[k0:k0+32].sum()),3))
print("lse sample", lse[0,0,:3], lse[0,0,-2:])
