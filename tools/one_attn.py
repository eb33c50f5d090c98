import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
dev='cuda'
which = sys.argv[1] if len(sys.argv) > 1 else 'attn'
if which == 'attn':
    B,H,N,Dh = 16,16,768,64
    q = torch.randn(B,H,N,Dh,device=dev).to(torch.bfloat16); k = torch.randn(B,H,N,Dh,device=dev).to(torch.bfloat16)
    vt = torch.randn(B,H,Dh,N,device=dev).to(torch.bfloat16); o = torch.empty(B,N,H*Dh,device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.attention(q,k,vt,o,B,H,N,N,N,N,Dh)
else:
    M,N,K = 12288,4096,1024
    x = torch.randn(M,K,device=dev).to(torch.bfloat16); w = (torch.randn(N,K,device=dev)*0.03).to(torch.bfloat16)
    out = torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(x,w,None,ops.EPI_BF16,out)
torch.cuda.synchronize()
