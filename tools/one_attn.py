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
elif which == 'render':
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import orbit_cameras
    tp = Triplane(img_resolution=256).to(dev)
    tp.decoder.net[2].bias.data[0] += 4.0
    pcl = torch.randn(1,3,128,128,32,device=dev)*4
    cams = orbit_cameras(4).to(dev); idx = torch.zeros(4,dtype=torch.int32,device=dev)
    j = torch.rand(4,65536,64,device=dev); u = torch.rand(4*65536,64,device=dev)
    for _ in range(2): tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=256, jitter=j, u_fine=u)
elif which == 'gemm_gelu':
    M,N,K = 12288,4096,1024
    x = torch.randn(M,K,device=dev).to(torch.bfloat16); w = (torch.randn(N,K,device=dev)*0.03).to(torch.bfloat16)
    b = torch.randn(N,device=dev)*0.02
    out = torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(x,w,b,ops.EPI_GELU_ERF,out)
else:
    M,N,K = 12288,4096,1024
    x = torch.randn(M,K,device=dev).to(torch.bfloat16); w = (torch.randn(N,K,device=dev)*0.03).to(torch.bfloat16)
    out = torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(x,w,None,ops.EPI_BF16,out)
torch.cuda.synchronize()
