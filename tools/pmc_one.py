"""One kernel at one shape, launched 3 times - the target of the rocprofv3 --pmc passes of tools/pmc_traffic.sh.
usage: pmc_one.py gemm M N K | gemm_gr M N K | attn BH Nq Nk Dh | render V RES"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
dev = 'cuda'
which, a = sys.argv[1], [int(v) for v in sys.argv[2:]]
if which == 'gemm':
    M, N, K = a
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(x, w, b, ops.EPI_GELU_ERF, out)
elif which == 'gemm_gr':                      # fc2: gate * out + residual into the fp32 stream (the kernel NAME with the largest share of a step)
    M, N, K = a
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    res = torch.randn(M, N, device=dev)
    gate = torch.randn(M // 768, 6 * N, device=dev) * 0.1
    for _ in range(3):
        ops.gemm(x, w, b, ops.EPI_GATE_RES, res, None, gate=gate, gate_rows=768, gate_ld=6 * N)
elif which == 'attn':
    BH, Nq, Nk, Dh = a
    H = 16
    B = BH // H
    q = torch.randn(B, H, Nq, Dh, device=dev).to(torch.bfloat16)
    k = torch.randn(B, H, Nk, Dh, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, H, Dh, Nk, device=dev).to(torch.bfloat16)
    o = torch.empty(B, Nq, H * Dh, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.attention(q, k, vt, o, B, H, Nq, Nq, Nk, Nk, Dh)
else:
    V, res = a
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import orbit_cameras
    tp = Triplane(img_resolution=128).to(dev)
    tp.decoder.net[2].bias.data[0] += 4.0
    pcl = torch.randn(1, 3, 128, 128, 32, device=dev) * 4
    cams = orbit_cameras(V).to(dev)
    idx = torch.zeros(V, dtype=torch.int32, device=dev)
    j = torch.rand(V, res * res, 64, device=dev)
    u = torch.rand(V * res * res, 64, device=dev)
    for _ in range(3):
        tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
torch.cuda.synchronize()
