import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
dev='cuda'
def run(B,H,Nq,Nk,Dh, mode='rand'):
    g = torch.Generator().manual_seed(1)
    nqp, nkp = (Nq+63)//64*64, (Nk+63)//64*64
    q = torch.zeros(B,H,nqp,Dh); k = torch.zeros(B,H,nkp,Dh); v = torch.zeros(B,H,nkp,Dh)
    q[:,:,:Nq] = torch.randn(B,H,Nq,Dh,generator=g); k[:,:,:Nk] = torch.randn(B,H,Nk,Dh,generator=g)
    if mode == 'vkey':   # v[key, d] = key index -> output = expected key under softmax
        v[:,:,:Nk] = torch.arange(Nk).float()[None,None,:,None].expand(B,H,Nk,Dh) / 64.0
    elif mode == 'vd':
        v[:,:,:Nk] = torch.arange(Dh).float()[None,None,None,:].expand(B,H,Nk,Dh) / 8.0
    else:
        v[:,:,:Nk] = torch.randn(B,H,Nk,Dh,generator=g)
    qb,kb,vb = (t.to(torch.bfloat16).to(dev) for t in (q,k,v))
    vt = vb.transpose(-1,-2)[..., ops.vt_key_order(nkp, dev)].contiguous()
    out = torch.empty(B,Nq,H*Dh,device=dev,dtype=torch.bfloat16)
    ops.attention(qb,kb,vt,out,B,H,Nq,nqp,Nk,nkp,Dh)
    s = (qb[:,:,:Nq].float() @ kb[:,:,:Nk].float().transpose(-1,-2)) * Dh**-0.5
    ref = (torch.softmax(s,-1) @ vb[:,:,:Nk].float()).permute(0,2,1,3).reshape(B,Nq,H*Dh)
    err = (out.float()-ref)
    rel = float(err.norm()/ref.norm())
    per_row = err.reshape(B,Nq,H,Dh).norm(dim=-1) / (ref.reshape(B,Nq,H,Dh).norm(dim=-1)+1e-9)
    bad = (per_row > 0.05)
    print(f'B{B} H{H} Nq{Nq} Nk{Nk} Dh{Dh} {mode}: rel {rel:.4f}; bad rows {int(bad.sum())}/{bad.numel()}', end='')
    if bad.any():
        idx = bad.nonzero()[:6].tolist()
        print(' first bad (b,q,h):', idx, end='')
    print()
    return out, ref
for Nk in (64, 128, 192, 256, 768):
    run(1,1,256,Nk,64)
run(1,1,256,256,64,'vkey'); run(1,1,256,256,64,'vd')
run(1,1,32,64,64); run(1,2,256,64,64); run(1,1,256,64,128); run(1,1,256,256,128)
o, r = run(1,1,32,64,64,'vkey')
print(o[0,:4,:8].float().cpu(), r[0,:4,:8].cpu())
