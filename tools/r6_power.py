#!/usr/bin/env python
"""r6: is the denoise loop paced by the part's power management?  Socket power / cap / shader clock / hotspot
sampled at 20 Hz by a side process (tools/power_sampler.py) while this process runs 4-second loops of single kernels (random and
zero-filled operands, the vendor GEMM as a yardstick) and then the bench's own step.  Output: gpurun_out/r6_power.csv (trace),
gpurun_out/r6_power_phases.json (phase boundaries + per-phase mean / max), a markdown table on stdout.  GPU box only."""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ln3diff_amd import ops  # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
csv = os.path.join(OUT, 'r6_power.csv')
stop = csv + '.stop'
for f in (csv, stop):
    if os.path.exists(f):
        os.remove(f)
sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'power_sampler.py'), csv, '20', stop],
                           stderr=open(os.path.join(OUT, 'r6_power_sampler.err'), 'w'))
dev = torch.device('cuda:0')
SEC = float(os.environ.get('R6_POWER_SEC', '4'))
phases = []


def loop(name, fn, sec=SEC, flops=0.0, bytes_=0.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < sec:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    us = e0.elapsed_time(e1) * 1e3 / n
    phases.append(dict(name=name, t0=t0, t1=t1, launches=n, avg_us=round(us, 2), tflops=round(flops / us / 1e6, 1) if flops else None,
                       tbps=round(bytes_ / us / 1e6, 2) if bytes_ else None))
    time.sleep(1.0)                                    # let the trace fall back between phases


def idle(name, sec):
    t0 = time.time()
    time.sleep(sec)
    phases.append(dict(name=name, t0=t0, t1=time.time()))


idle('idle', 3.0)
M = 16 * 768
xr = torch.randn(M, 1024, device=dev).to(torch.bfloat16)
wr = (torch.randn(4096, 1024, device=dev) * 0.03).to(torch.bfloat16)
b4 = torch.randn(4096, device=dev) * 0.02
y = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
fl = 2.0 * M * 4096 * 1024
loop('fc1+GELU shipped, random operands', lambda: ops.gemm(xr, wr, b4, ops.EPI_GELU_ERF, y), flops=fl)
xz, wz = torch.zeros_like(xr), torch.zeros_like(wr)
loop('fc1+GELU shipped, ZERO operands', lambda: ops.gemm(xz, wz, b4, ops.EPI_GELU_ERF, y), flops=fl)
loop('fc1 shape plain bf16 shipped, random', lambda: ops.gemm(xr, wr, b4, ops.EPI_BF16, y), flops=fl)
wrt = wr.t()
loop('fc1 shape VENDOR (torch.matmul), random', lambda: torch.matmul(xr, wrt, out=y), flops=fl)
wzt = wz.t()
loop('fc1 shape VENDOR (torch.matmul), ZERO', lambda: torch.matmul(xz, wzt, out=y), flops=fl)
x2 = torch.randn(M, 4096, device=dev).to(torch.bfloat16)
w2 = (torch.randn(1024, 4096, device=dev) * 0.02).to(torch.bfloat16)
b1 = torch.randn(1024, device=dev) * 0.02
res = torch.zeros(M, 1024, device=dev)
gate = torch.randn(16, 6 * 1024, device=dev) * 0.1
loop('fc2 gate/residual shipped, random', lambda: ops.gemm(x2, w2, b1, ops.EPI_GATE_RES, res, None, gate=gate, gate_rows=768, gate_ld=6144),
     flops=2.0 * M * 1024 * 4096)
res.zero_()
loop('attention out-proj gate/residual (K=1024)', lambda: ops.gemm(xr, w2[:, :1024].contiguous(), b1, ops.EPI_GATE_RES, res, None, gate=gate, gate_rows=768, gate_ld=6144),
     flops=2.0 * M * 1024 * 1024, bytes_=M * 1024 * (2 + 8) + 2 * 1024 * 1024)
B, H, N, Dh = 16, 16, 768, 64
q = torch.randn(B, H, N, Dh, device=dev).to(torch.bfloat16)
k = torch.randn(B, H, N, Dh, device=dev).to(torch.bfloat16)
vt = torch.randn(B, H, Dh, N, device=dev).to(torch.bfloat16)
o = torch.empty(B, N, H * Dh, device=dev, dtype=torch.bfloat16)
loop('self-attention 256 heads x 768^2 x 64', lambda: ops.attention(q, k, vt, o, B, H, N, N, N, N, Dh), flops=4.0 * N * N * H * Dh * B)
xf = torch.randn(M, 1024, device=dev)
yb = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
mod = torch.randn(16, 6 * 1024, device=dev)
loop('norm_modulate 12288 x 1024', lambda: ops.norm_modulate(xf, yb, M, 1024, shift=mod, scale=mod[:, 1024:], mod_rows=768, mod_ld=6144), bytes_=M * 1024 * 6.0)
del xr, wr, y, x2, w2, res, q, k, vt, o, xf, yb
torch.cuda.empty_cache()

# the bench's own step (1 warm-up + 2 timed batches of 8: 250-step denoise loop, decode, 320 views), in a child process
t0 = time.time()
p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-probes', '--unfolded-steps', '0'],
                   capture_output=True, text=True)
t1 = time.time()
line = [l for l in p.stdout.splitlines() if l.startswith('{')]
bench = json.loads(line[-1]) if line else {'error': p.stderr[-2000:]}
phases.append(dict(name='bench.py configs[1] (1 warm-up + 2 timed steps; includes model build)', t0=t0, t1=t1, value=bench.get('value'), ms_per_step=bench.get('ms_per_step')))
idle('idle after', 2.0)
open(stop, 'w').close()
sampler.wait(timeout=20)

rows = []
src = '?'
for l in open(csv):
    if l.startswith('# source='):
        src = l.strip()[9:]
    if l.startswith('#') or l.startswith('t,'):
        continue
    v = [float(a) for a in l.strip().split(',')]
    rows.append(v)
cols = ['t', 'power_w', 'cap_w', 'sclk_mhz', 'sclk_min', 'sclk_max', 'hotspot_c', 'mem_c', 'uclk_mhz', 'throttle', 'gfx_busy', 'energy']
ix = {c: i for i, c in enumerate(cols)}


def stat(ph, c, skip=0.7):
    # samples inside the phase, the first `skip` seconds dropped (the SMU's averaging window and the ramp)
    v = [r[ix[c]] for r in rows if ph['t0'] + skip <= r[0] <= ph['t1'] and r[ix[c]] == r[ix[c]]]
    return (sum(v) / len(v), max(v), min(v), len(v)) if v else (float('nan'),) * 3 + (0,)


print('source: %s, %d samples' % (src, len(rows)))
print('| phase | launches | avg us | TFLOP/s or TB/s | power W mean / max | cap W | sclk MHz mean / min | hotspot C | throttle | samples |')
print('|---|---|---|---|---|---|---|---|---|---|')
for ph in phases:
    pw, sc, hs, cp, th = stat(ph, 'power_w'), stat(ph, 'sclk_mhz'), stat(ph, 'hotspot_c'), stat(ph, 'cap_w'), stat(ph, 'throttle')
    ph.update(power_w_mean=round(pw[0], 1), power_w_max=round(pw[1], 1), cap_w=round(cp[0], 1), sclk_mean=round(sc[0], 0), sclk_min=round(sc[2], 0),
              hotspot_c=round(hs[1], 1), throttle_max=th[1], samples=pw[3])
    print('| %s | %s | %s | %s | %.0f / %.0f | %.0f | %.0f / %.0f | %.0f | %s | %d |' % (
        ph['name'], ph.get('launches', ''), ph.get('avg_us', ''), ph.get('tflops') or ph.get('tbps') or ph.get('value', ''), pw[0], pw[1], cp[0], sc[0], sc[2], hs[1], th[1], pw[3]))
json.dump(dict(source=src, phases=phases, bench=bench), open(os.path.join(OUT, 'r6_power_phases.json'), 'w'), indent=1)
# the bench phase as a coarse time series (0.5 s bins) so that denoise loop vs decode + render is visible
bp = phases[-2]
print('\nbench phase, 0.5 s bins: t (s since phase start), power W, sclk MHz')
tb = bp['t0']
while tb < bp['t1']:
    v = [(r[ix['power_w']], r[ix['sclk_mhz']]) for r in rows if tb <= r[0] < tb + 0.5]
    if v:
        print('%6.1f  %6.0f  %6.0f' % (tb - bp['t0'], sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
    tb += 0.5
