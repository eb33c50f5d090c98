import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
torch.manual_seed(0)
dev='cuda'; Bn,N,K,H=2,768,1024,16; Lc=int(sys.argv[2]) if len(sys.argv)>2 else 77; M,D=Bn*N,H*64; lpad=128
x=torch.randn(M,K,device=dev).to(torch.bfloat16); w=(torch.randn(D,K,device=dev)*0.03).to(torch.bfloat16)
kc=torch.zeros(Bn,H,lpad,64,device=dev,dtype=torch.bfloat16); vc=torch.zeros_like(kc)
kc[:,:,:Lc]=torch.randn(Bn,H,Lc,64,device=dev).to(torch.bfloat16); vc[:,:,:Lc]=torch.randn(Bn,H,Lc,64,device=dev).to(torch.bfloat16)
mode=sys.argv[1] if len(sys.argv)>1 else 'full'
if mode=='vconst':   # V = 1 everywhere -> output must be 1
    vc[:,:,:Lc]=1.0
if mode=='kzero':    # K = 0 -> uniform softmax -> output = mean of V over keys
    kc.zero_()
vt=vc.transpose(-1,-2).contiguous()[..., ops.vt_key_order(lpad,dev)].contiguous()
kp=kc[..., ops.vt_key_order(64,dev)].contiguous()
out=torch.empty(M,D,device=dev,dtype=torch.bfloat16)
ops.gemm(x,w,None,ops.EPI_CROSS_ATTN,out,kp,vt,M=M,tokens=N,heads=H,head_dim=64,ctx_keys=Lc,ctx_pad=lpad,ctx_scale=0.125)
q=(x.float()@w.float().t()).to(torch.bfloat16).float().view(Bn,N,H,64).transpose(1,2)
a=torch.softmax(q@kc[:,:,:Lc].float().transpose(-1,-2)*0.125,-1)@vc[:,:,:Lc].float()
ref=a.transpose(1,2).reshape(M,D)
o=out.float()
print(mode,'rel',float((o-ref).norm()/ref.norm()),'finite',bool(torch.isfinite(o).all()))
print('out[0,:8]',o[0,:8].tolist()); print('ref[0,:8]',ref[0,:8].tolist())
print('out[5,64:72]',o[5,64:72].tolist()); print('ref[5,64:72]',ref[5,64:72].tolist())
for h in range(0,H,5):
    e=float((o[:,h*64:(h+1)*64]-ref[:,h*64:(h+1)*64]).norm()/ref[:,h*64:(h+1)*64].norm()); print('head',h,e)
