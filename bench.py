#!/usr/bin/env python
"""bench.py - 3D samples/sec of the text->3D sampling hot path on MI355X (driver contract in the task brief).

One "step" = one pass of the whole hot path over one batch of B samples per GPU:
  noise z[B,12,32,32] -> EulerEDM x 250 steps with CFG 6.5 (500 DiT-L/2 forwards per sample; LegacyDDPM sigmas,
  the released T23D sampler) -> latent*0.96806 -> VAE decode (DiT2-L/2 + conv decoder) -> V views @ res^2 through
  the fused tri-plane ray-marcher (64+64 samples/ray).  Weights: random-init of the named architectures; inputs:
  synthetic noise / conditioning / orbit cameras, resident in HBM before the timed region.
N GPUs: one process per GPU, weak scaling (B samples per rank), RCCL broadcast of rank 0's weights at start-up,
all_gather of the final latents inside the timed region; no collective inside the denoise loop.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _sha16(rel):
    import hashlib
    with open(os.path.join(ROOT, rel), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(key):
    """HBM / fabric bytes per launch of a kernel at a shape, from the committed rocprofv3 --pmc passes (profiles/r6_pmc.json:
    FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes; bench.py cannot run the profiler on itself).  An entry is
    only valid for the kernel source it was measured on: it carries the sha256 of the .hip file, and a different source on disk
    yields null plus the reason instead of a stale number."""
    path = os.path.join(ROOT, 'profiles', 'r6_pmc.json')
    if not os.path.exists(path):
        return None, 'profiles/r6_pmc.json missing'
    ent = json.load(open(path)).get(key)
    if ent is None:
        return None, 'no PMC entry for %s in profiles/r6_pmc.json' % key
    cur = _sha16(ent['hip'])
    if cur != ent['sha16']:
        return None, '%s changed since the PMC pass (sha %s, measured on %s): re-run tools/r6_pmc.sh' % (ent['hip'], cur, ent['sha16'])
    return ent['traffic_bytes'], ent['source']


def build_models(dev, arch, dec_arch, seed=0, fill=True, golden_weights=False):
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.dit.dit_decoder import DiT2_models
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.vit.vit_triplane import (
        RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder as Dec)
    from ln3diff_amd.synth import fill_module_random_
    dit = DiT_models[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768,
                           roll_out=True, vit_blk=TextCondDiTBlock)
    vit = DiT2_models[dec_arch](input_size=16, num_classes=0, learn_sigma=False, in_channels=dit.embed_dim,
                                mixed_prediction=False, context_dim=None, roll_out=True, plane_n=3)
    dec = Dec(vit_decoder=vit, triplane_decoder=Triplane(img_resolution=128), cls_token=False, vae_p=2,
              ldm_z_channels=4, ldm_embed_dim=4)
    dit, dec = dit.to(dev), dec.to(dev)
    if fill:
        if golden_weights:            # the (name, shape, seed) weights of tests/golden: lets the timed run be checked against a golden
            from ln3diff_amd.synth import load_synth_
            load_synth_(dit, seed)
            load_synth_(dec, seed + 1)   # r6: the decoder too, so that the timed run's PICTURE can be checked (golden_check, full_chain_ditl2)
        else:
            fill_module_random_(dit, seed, dev)
            fill_module_random_(dec, seed + 1, dev)
        # keep the synthetic volume non-empty so compositing is exercised
        dec.triplane_decoder.decoder.net[2].bias.data[0] += 4.0
    return dit, dec


class ClockSampler:
    """Shader clock / socket power during a region, sampled by a thread through amdsmi's gpu_metrics (rank 0 only, ~10 Hz, read-only;
    None values when the library is absent).  r6: the GEMMs run AT the 1 400 W cap with the clock pulled to 1.74 - 1.9 GHz
    (profiles/r6_power.md), so every roofline fraction is printed with the clock it was measured at."""

    def __init__(self, hz=20.0):
        import threading
        self.rows, self._stop, self.cap = [], threading.Event(), None
        self._thr = threading.Thread(target=self._run, args=(hz,), daemon=True)
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self._h, self._smi = amdsmi.amdsmi_get_processor_handles()[0], amdsmi
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self._h).get('power_cap')
                self.cap = cap / 1e6 if isinstance(cap, (int, float)) and cap > 100000 else cap
            except Exception:                                   # noqa: BLE001
                pass
        except Exception:                                       # noqa: BLE001
            self._smi = None

    def _run(self, hz):
        while not self._stop.is_set():
            try:
                m = self._smi.amdsmi_get_gpu_metrics_info(self._h)
                clk = [c for c in (m.get('current_gfxclks') or []) if isinstance(c, (int, float)) and 0 < c < 60000]
                pw = m.get('current_socket_power')
                if clk and isinstance(pw, (int, float)):
                    self.rows.append((time.time(), sum(clk) / len(clk), float(pw)))
            except Exception:                                   # noqa: BLE001
                pass
            self._stop.wait(1.0 / hz)

    def __enter__(self):
        if self._smi is not None:
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._smi is not None:
            self._thr.join(timeout=2.0)

    def summary(self, t0=None, t1=None, busy_w=600.0):
        """Mean clock / power over [t0, t1]; *_busy = over the samples above busy_w watts (the denoise loop and the GEMM probes sit at
        1.2 - 1.4 kW, idle at 0.25 - 0.3 kW)."""
        r = [x for x in self.rows if (t0 is None or x[0] >= t0) and (t1 is None or x[0] <= t1)]
        if not r:
            return None
        b = [x for x in r if x[2] >= busy_w] or r
        return {"sclk_mhz": round(sum(x[1] for x in b) / len(b)), "power_w": round(sum(x[2] for x in b) / len(b)), "power_cap_w": self.cap,
                "samples": len(b), "source": "amdsmi gpu_metrics current_gfxclks / current_socket_power, samples above %d W" % busy_w}


def _with_clock(rec, clk, t0, t1, nominal_mhz=2400.0):
    """sclk_mhz / power_w of the probe's own loop and frac_at_clock = achieved / (peak x sclk / 2400 MHz) beside the datasheet fraction."""
    cs = clk.summary(t0, t1) if clk is not None else None
    if cs and rec.get("frac") is not None and rec.get("bound") == "mfma":
        rec["sclk_mhz"], rec["power_w"], rec["power_cap_w"] = cs["sclk_mhz"], cs["power_w"], cs["power_cap_w"]
        rec["frac_at_clock"] = round(rec["achieved"] / (rec["peak"] * cs["sclk_mhz"] / nominal_mhz), 4)
    elif cs:
        rec["sclk_mhz"], rec["power_w"], rec["power_cap_w"] = cs["sclk_mhz"], cs["power_w"], cs["power_cap_w"]
    return rec


def mfma_probe(dev, seconds=1.0):
    """What the matrix pipes of THIS box sustain on a pure-MFMA stream (ln3d_probe_mfma_bf16: no memory traffic, 2 x 256 workgroups x 8
    waves), ~1 s of back-to-back launches.  Printed beside the datasheet peak; `frac` stays achieved / datasheet (the contract's peak and
    comparable across rounds), `frac_of_sustained` = achieved / this."""
    from ln3diff_amd import ops
    wgs, iters = 512, 20000
    out = torch.empty(wgs * 512, device=dev)
    ops.probe_mfma(out, wgs, 200)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    n, fl = 0, 0.0
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(4):
            fl += ops.probe_mfma(out, wgs, iters)
            n += 1
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return {"tflops": round(fl / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1), "launches": n, "t0": t0, "t1": time.time(),
            "what": "ln3d_probe_mfma_bf16: 512 workgroups x 8 waves x 20000 x 8 v_mfma_f32_32x32x16_bf16, no memory traffic"}


def _timed_loop(f, iters, min_seconds=0.6):
    """Average launch time (ms) over back-to-back launches on the current stream, HIP events around the whole loop; batches of `iters`
    until `min_seconds` have passed, so that the side-thread clock sampler sees the loop and the part reaches its sustained clock."""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    while True:
        for _ in range(iters):
            f()
        n += iters
        torch.cuda.synchronize()
        if time.time() - t0 >= min_seconds:
            break
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def roofline_probe(dev, n_net, D, iters=20, tokens=768):
    """Live HIP-event timing of the dominant kernel of the step: the MLP fc1 GEMM (gemm_bf16_kernel<GELU_ERF>,
    M = n_net*768 tokens, N = 4D, K = D) at the workload's exact shape, on the stream it is launched on."""
    from ln3diff_amd import ops
    M, N, K = n_net * tokens, 4 * D, D
    x = (torch.randn(M, K, device=dev) * 1.0).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = _timed_loop(lambda: ops.gemm(x, w, b, ops.EPI_GELU_ERF, y), iters)
    flops = 2.0 * M * N * K
    peak = 2500.0   # TFLOP/s dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = pmc_traffic("gemm_fc1_gelu_%dx%dx%d" % (M, N, K))
    return {"kernel": "gemm_bf16_p4_kernel<GELU_ERF> (DiT MLP fc1: persistent 256x256 tiles, one wave per SIMD; r5: gemm_bf16_ring64_kernel<GELU_ERF, 256x256>)", "shape": [M, N, K], "bound": "mfma", "achieved": round(ach, 1),
            "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": src, "avg_us": round(ms * 1e3, 2),
            "algorithmic_flop_per_launch": flops}


def gate_res_probe(dev, n_net, D, iters=20, tokens=768):
    """The kernel NAME with the largest share of the step is the gate / residual GEMM (attention projection + MLP fc2 share one
    instantiation): probe of its bigger member, fc2 (M = n_net * tokens, N = D, K = 4D), fp32 residual stream updated in place."""
    from ln3diff_amd import ops
    M, N, K = n_net * tokens, D, 4 * D
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    res = torch.randn(M, N, device=dev)
    gate = torch.randn(n_net, 6 * N, device=dev) * 0.1
    f = lambda: ops.gemm(x, w, b, ops.EPI_GATE_RES, res, None, gate=gate, gate_rows=tokens, gate_ld=6 * N)
    ms = _timed_loop(f, iters)
    flops = 2.0 * M * N * K
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_bf16_ring64_kernel<GATE_RES, 256x192> (DiT MLP fc2: gate * out + residual into the fp32 stream)", "shape": [M, N, K],
            "bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "avg_us": round(ms * 1e3, 2),
            "traffic": pmc_traffic("gemm_fc2_gateres_%dx%dx%d" % (M, N, K))[0], "traffic_source": pmc_traffic("gemm_fc2_gateres_%dx%dx%d" % (M, N, K))[1],
            "algorithmic_bytes": "X %d MB + W %d MB read, %d MB residual read + written" % (M * K * 2 // 2 ** 20, N * K * 2 // 2 ** 20, M * N * 8 // 2 ** 20)}


def out_proj_probe(dev, n_net, D, iters=40, tokens=768):
    """The N = K = D members of the gate / residual family (attention out-proj; cross-attention to_out at half the rows): 25.8 GFLOP behind
    a 100 MB fp32 residual read-modify-write - 203 flop per byte is UNDER the part's ridge (2500 / 8 = 312), so the bound is HBM / fabric
    traffic (they were graded against the MFMA peak).  achieved = algorithmic bytes / time."""
    from ln3diff_amd import ops
    M, N, K = n_net * tokens, D, D
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    res = torch.randn(M, N, device=dev)
    gate = torch.randn(n_net, 6 * N, device=dev) * 0.1
    f = lambda: ops.gemm(x, w, b, ops.EPI_GATE_RES, res, None, gate=gate, gate_rows=tokens, gate_ld=6 * N)
    ms = _timed_loop(f, iters)
    byts = M * K * 2.0 + N * K * 2.0 + M * N * 8.0
    return {"kernel": "gemm_bf16_ring64_kernel<GATE_RES> at N = K = %d (attention out-proj: gate * out + residual into the fp32 stream)" % D,
            "shape": [M, N, K], "bound": "hbm", "achieved": round(byts / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(byts / (ms * 1e-3) / 1e9 / 8000.0, 4), "avg_us": round(ms * 1e3, 2), "traffic": None,
            "algorithmic_bytes": "X %.1f MB + W %.1f MB read, %.1f MB residual read + written" % (M * K * 2 / 2 ** 20, N * K * 2 / 2 ** 20, M * N * 8 / 2 ** 20),
            "mfma_frac_for_reference": round(2.0 * M * N * K / (ms * 1e-3) / 1e12 / 2500.0, 4)}


def attention_probe(dev, n_net, H=16, N=768, Dh=64, iters=20, Nq=None):
    from ln3diff_amd import ops
    Nq = Nq or N
    Dh_true = Dh
    from ln3diff_amd.dit.dit_models_xformers import attn_head_pad
    Dh = attn_head_pad(Dh)                         # as the model stores them: DiT-XL/2's 72-wide heads sit in zero-padded 80-wide rows (r6; 128 before)
    q = torch.randn(n_net, H, Nq, Dh, device=dev).to(torch.bfloat16)
    k = torch.randn(n_net, H, N, Dh, device=dev).to(torch.bfloat16)
    vt = torch.randn(n_net, H, Dh, N, device=dev).to(torch.bfloat16)
    dt = Dh_true if Dh_true != Dh else 0            # as the model calls it: true head size, compact [.., H * Dh_true] output rows
    if dt:
        q[..., dt:] = 0; k[..., dt:] = 0; vt[:, :, dt:, :] = 0
    o = torch.empty(n_net, Nq, H * (dt or Dh), device=dev, dtype=torch.bfloat16)
    ms = _timed_loop(lambda: ops.attention(q, k, vt, o, n_net, H, Nq, Nq, N, N, Dh, scale=Dh_true ** -0.5, dh_true=dt), iters)
    flops = 4.0 * Nq * N * H * Dh_true * n_net     # 4*Nq*Nkv*(H*Dh) per sample-layer (algorithmic: the true head size)
    ach = flops / (ms * 1e-3) / 1e12
    kres = N % 256 == 0 and 512 <= N <= 768 and Nq % 256 == 0 and Dh == 64 and n_net * H >= 256
    stream = N % 256 == 0 and Dh == 64 and not kres
    name = ("attn_kres_kernel (K resident in LDS, V^T ring, row sums on the matrix pipe)" if kres else
            "attn_stream_kernel (one workgroup per head, 8-slot LDS-DMA K/V ring)" if stream else "attn_kernel<%d, DT %d> (tiled ring kernel; %d-wide heads in %d-wide rows, the padding skipped)" % (Dh, Dh_true, Dh_true, Dh))
    traffic, src = pmc_traffic("attention_%dx%dx%dx%d" % (n_net * H, Nq, N, Dh))
    return {"kernel": name + " - DiT self-attention, %d queries x %d keys" % (Nq, N), "bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0,
            "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "avg_us": round(ms * 1e3, 2), "traffic": traffic, "traffic_source": src}


def pmc_entry(key):
    """The whole committed PMC record of a kernel (profiles/r6_pmc.json), or None when absent / measured on another source."""
    path = os.path.join(ROOT, 'profiles', 'r6_pmc.json')
    if not os.path.exists(path):
        return None
    ent = json.load(open(path)).get(key)
    return ent if ent is not None and _sha16(ent['hip']) == ent['sha16'] else None


def render_probe(dev, dec, res=256, V=4, iters=5):
    """The fused ray-marcher.  It is NOT an HBM-streaming kernel: one tri-plane is 6 MB and stays in L2 (fabric traffic ~100 MB per
    256^2 view against 12.9 GB of gathered texel bytes), so SURVEY 8d's "gather bytes / HBM peak" is not a roofline for it (r1 - r4
    printed that ratio as frac = 2.2).  What bounds it is vector-instruction issue: `bound` = "valu-issue", `frac` = the SIMDs'
    VALU-busy cycles / elapsed cycles from the committed counter pass (SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs),
    profiles/r6_pmc.json, tools/r6_pmc.sh), `achieved` / `peak` = vector instructions per second issued / issuable (1024 SIMDs x
    clock / 4 cycles per wave64 instruction at the measured clock).  The texel-gather rate is kept as l2_gather_GBps."""
    from ln3diff_amd.synth import orbit_cameras
    tp = dec.triplane_decoder
    pcl = torch.randn(1, 3, 128, 128, 32, device=dev) * 4.0
    cams = orbit_cameras(V).to(dev)
    idx = torch.zeros(V, dtype=torch.int32, device=dev)
    j = torch.rand(V, res * res, 64, device=dev)
    u = torch.rand(V * res * res, 64, device=dev)
    tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pts = V * res * res * 128
    gbytes = pts * 1536.0 / 1e9                    # 3 planes x 4 taps x 32 ch x 4 B per sample point
    mlp_tflops = pts * 2.0 * (32 * 64 + 64 * 4) * 3 / (ms * 1e-3) / 1e12      # bf16x3 split: 3 MFMA products per fp32 product
    rec = {"kernel": "render_kernel (fused tri-plane ray-march, %dx%d^2 views)" % (V, res), "bound": "valu-issue", "unit": "G vector instructions/s",
           "ms_per_view": round(ms / V, 3), "target_ms_per_view": 2.7, "l2_gather_GBps": round(gbytes / (ms * 1e-3), 1),
           "mlp_issued_bf16_tflops": round(mlp_tflops, 1),
           "traffic": pmc_traffic("render_%dx%d" % (V, res))[0], "traffic_source": pmc_traffic("render_%dx%d" % (V, res))[1]}
    ent = pmc_entry("render_%dx%d" % (V, res))
    if ent and ent.get("issue"):
        iss = ent["issue"]
        clock_ghz = iss["cycles_per_launch"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0          # counter cycles over this run's launch time
        rec.update({"achieved": round(iss["valu_insts"] / (ms * 1e-3) / 1e9, 1), "peak": round(1024 * clock_ghz / 4, 1),
                    "frac": round(iss["valu_busy_frac"], 4), "valu_insts_per_ray": round(iss["valu_insts"] / (V * res * res), 0),
                    "mfma_busy_frac": round(iss["mfma_busy_frac"], 4), "lds_busy_frac": round(iss["lds_busy_frac"], 4),
                    # r6: the other throughput term of the kernel (profiles/r6_render_abl.log: without the texel loads it is 30 % faster) - every
                    # vector-memory read of the marcher is a wave64 dwordx4 = 1 KB through the CU's L1 path, taken at 64 B per clock
                    "l1_path_busy_est": (round(ent["counters"]["SQ_INSTS_VMEM_RD"] * 16.0 / (iss["cycles_per_launch"] * 256.0), 4)
                                         if ent.get("counters", {}).get("SQ_INSTS_VMEM_RD") else None),
                    "issue_source": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (tools/r6_pmc.sh; "
                                    "quad-cycle counters x 4, 1024 SIMDs); frac = VALU-busy SIMD cycles / elapsed SIMD cycles"})
    else:
        rec.update({"achieved": None, "peak": None, "frac": None, "issue_source": "no counter pass for this source (profiles/r6_pmc.json): tools/r6_pmc.sh"})
    return rec


def _pick_threads():
    """"all host cores, count stated".  torch's CPU kernels get SLOWER past the socket's sweet spot on the 256-thread GPU
    hosts (141 s per DiT-L/2 step with 256 threads vs ~2.4 s with 32), so three thread counts are probed on a GEMM + GELU of the
    workload's shape, the best one is used, and all three timings are reported."""
    host = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    xa, wa = torch.randn(1536, 1024), torch.randn(4096, 1024)
    probe = {}
    for n in sorted({max(1, min(host, c)) for c in (16, 32, 64)}):
        torch.set_num_threads(n)
        torch.nn.functional.linear(xa, wa)
        t0 = time.time()
        for _ in range(5):
            torch.nn.functional.gelu(torch.nn.functional.linear(xa, wa))
        probe[n] = round((time.time() - t0) / 5 * 1e3, 2)
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    return cores, host, probe


def _cpu_view(res_target, g):
    """One view of the oracle's renderer at 64^2, scaled by the ray count (the cost is linear in rays)."""
    from oracle import render as orender
    from ln3diff_amd.synth import orbit_cameras
    dec_sd = {'net.0.weight': torch.randn(64, 32, generator=g), 'net.0.bias': torch.zeros(64),
              'net.2.weight': torch.randn(4, 64, generator=g), 'net.2.bias': torch.tensor([4., 0, 0, 0])}
    planes = torch.randn(1, 96, 128, 128, generator=g) * 4
    rr = min(res_target, 64)
    jit = torch.rand(1, rr * rr, 64, 1, generator=g)
    uf = torch.rand(rr * rr, 64, generator=g)
    t0 = time.time()
    orender.triplane_render(planes, dec_sd, orbit_cameras(1), rr, jit, uf)
    return (time.time() - t0) * (res_target / rr) ** 2, rr


def cpu_baseline(arch, steps_total, views, res, B, i23d=False):
    """CPU restatement (oracle/, validated against the reference's own Python in the build container) timed on this box's host
    cores on a bounded sample of the same workload and extrapolated linearly (per-step and per-view costs are constant, BASELINE.md
    2): 1 warm-up + 5 timed network evaluations at B = 1 with CFG (network batch 2) - cut to 2 when a step exceeds 6 s, so the
    default run stays within its few minutes -, the VAE decode scaled from the step by FLOPs, one view at 64^2 scaled to the
    target resolution by rays."""
    from oracle import dit as odit, samplers as osamp
    cores, host, probe = _pick_threads()
    t_all = time.time()
    g = torch.Generator().manual_seed(0)
    hidden, depth, heads = odit.DIT_CONFIGS['DiT-L/2' if i23d else arch]
    if i23d:
        from ln3diff_amd.dit.dit_i23d import DiT_models
        m = DiT_models[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=1024, roll_out=True,
                             pooling_ctx_dim=768)
    else:
        from ln3diff_amd.dit.dit_trilatent import DiT_models
        from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
        m = DiT_models[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768, roll_out=True,
                             vit_blk=TextCondDiTBlock)
    sd = {k: v for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if 'pos_embed' not in k and ('adaLN' in k or 'final_layer' in k or 'scale_shift_table' in k):
            v.copy_(torch.randn(v.shape, generator=g) * 0.02)
    z = torch.randn(1, 12, 32, 32, generator=g)
    if i23d:
        zs = torch.cat([z, z])
        ctx = {'crossattn': torch.cat([torch.randn(1, 256, 2048, generator=g), torch.zeros(1, 256, 2048)]),
               'vector': torch.cat([torch.randn(1, 768, generator=g), torch.zeros(1, 768)])}
        step = lambda i: odit.i23d_forward_with_cfg(sd, zs, torch.full((2,), i / 49.0), ctx, 4.0, heads)
        n_eval, flops_step = steps_total - 1, 2 * 745.0
    else:
        cond = {'crossattn': torch.randn(1, 77, 768, generator=g)}
        uc = {'crossattn': torch.zeros(1, 77, 768)}
        table = osamp.discrete_denoiser_table()
        sig = osamp.legacy_ddpm_sigmas(250)
        net = lambda x, t, c: odit.t23d_forward(sd, x, t, c, heads)
        step = lambda i: osamp.edm_denoise_cfg(net, z, sig[i:i + 1], cond, uc, 6.5, table)
        n_eval, flops_step = steps_total, 2 * {'DiT-B/2': 178.0, 'DiT-L/2': 613.0, 'DiT-XL/2': 879.0}.get(arch, 613.0)
    with torch.no_grad():
        t0 = time.time()
        step(0)                                                             # warm-up
        t_warm = time.time() - t0
        n_timed = 5 if t_warm < 6.0 else 2
        t0 = time.time()
        for i in range(n_timed):
            step(i + 1)
        t_step = (time.time() - t0) / n_timed
        t_view, rr = _cpu_view(res, g)
    t_dec = t_step * (734.0 + 20.0) / flops_step                            # VAE decode: DiT2-L/2 (734 GFLOP) + conv decoder (~20)
    per_sample = n_eval * t_step + t_dec + views * t_view
    return {"value": round(1.0 / per_sample, 6), "unit": "3D samples/s", "cores": cores, "host_cores": host, "kind": "port",
            "thread_probe_ms": {str(k): v for k, v in probe.items()},
            "sample": "oracle/ (CPU restatement, fp32 torch, %d threads = the fastest of the 3-point probe): 1 warm-up + %d timed %s at B=1 "
                      "(%.2f s each) x %d, VAE decode scaled by FLOPs (%.2f s), 1 view at %d^2 scaled to %d^2 (%.2f s/view) x %d views; wall %.0f s"
                      % (cores, n_timed, "forward_with_cfg evaluations" if i23d else "EulerEDM+CFG steps", t_step, n_eval, t_dec, rr, res, t_view,
                         views, time.time() - t_all)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU over RCCL (the same
    command line the driver uses for N > 1).  Fails loudly when the box has fewer GPUs than requested."""
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit("bench.py --gpus %d: this box exposes %d GPU(s); refusing to report a %d-GPU number from fewer devices" % (n, have, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def build_i23d(dev, arch, seed=0, fill=True, golden_weights=False):
    from ln3diff_amd.dit.dit_i23d import DiT_models as I23D
    from ln3diff_amd.synth import fill_module_random_, load_synth_
    dit = I23D[arch](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=1024, roll_out=True,
                     pooling_ctx_dim=768).to(dev)
    if fill:
        if golden_weights:
            load_synth_(dit, seed)
        else:
            fill_module_random_(dit, seed, dev)
    return dit


def picture_check(latent0, rec_model, dec_arch):
    """r6: the PICTURE of global sample 0 of the last timed step against the reference's (tests/golden/full_chain_ditl2.npz, made by
    tests/golden/make_golden_full.py full_chain from the reference's own 250-step latent through the reference's AE + Triplane.forward):
    sample 0's latent is decoded and view 0 of the 40-camera orbit re-rendered at 256^2 AFTER the timed region with the fixture's render
    noise (stream seeded 0) and its triplane_scaling_divider 0.05 (at the released 0.96806 the random-init DiT's latent, std 18.7,
    saturates the decoder and the fp32 reference and the fp32 oracle differ by 6 - 8 % themselves: nothing to pin there)."""
    import numpy as np
    from ln3diff_amd.nsr.triplane import draw_render_noise
    from ln3diff_amd.pipeline import render_video_given_triplane
    path = os.path.join(ROOT, 'tests', 'golden', 'full_chain_ditl2.npz')
    if dec_arch != 'DiT2-L/2' or not os.path.exists(path):
        return None
    g = np.load(path)
    dev = latent0.device
    cams = torch.from_numpy(g['cams'][0:1]).to(dev)
    j, u = draw_render_noise(1, 256 * 256, 64, generator=torch.Generator().manual_seed(int(g['jitter_seed'])))
    out = render_video_given_triplane(latent0[None].clone().float(), rec_model, cams, triplane_scaling_divider=float(g['divider']), jitter=j, u_fine=u,
                                      resolution=256)
    st = int(g['stride'])
    rel = lambda a, b: float((a.double().cpu() - torch.from_numpy(b.astype(np.float64))).norm() / torch.from_numpy(b.astype(np.float64)).norm())
    return {"rgb_rel_l2": round(rel(out['image_raw'][0, 0][:, ::st, ::st], g['image_raw_sub'][0]), 6),
            "depth_rel_l2": round(rel(out['image_depth'][0, 0][:, ::st, ::st], g['image_depth_sub'][0]), 6),
            "planes_rel_l2": round(rel(out['latent_after_vit'][0][:, ::8, ::8], g['planes_sub'][0]), 6),
            "picture_fixture": "tests/golden/full_chain_ditl2.npz (view 0 of the orbit @ 256^2, divider 0.05, re-rendered after the timed region)"}


def golden_check(latent0, i23d, arch, sample_steps):
    """Sample 0 of the timed run (global sample 0: the golden's inputs, weights and sampler settings) against the final latent of
    the reference's own B = 1 loop (tests/golden/full_edm_ditl2_250.npz / full_flow_pixartl2_euler50.npz, made by
    tests/golden/make_golden_full.py in the build container): what ran at the benchmarked batch geometry IS the reference's
    computation, not merely something finite.  configs[1], configs[2] and (r4) configs[3]'s DiT-XL/2 have a fixture."""
    import numpy as np
    name = ('full_flow_pixartl2_euler50' if (i23d and arch == 'DiT-PixArt-L/2' and sample_steps == 50) else
            'full_edm_ditl2_250' if (not i23d and arch == 'DiT-L/2' and sample_steps == 250) else
            'full_edm_ditxl2_250' if (not i23d and arch == 'DiT-XL/2' and sample_steps == 250) else None)
    path = os.path.join(ROOT, 'tests', 'golden', (name or '') + '.npz')
    if name is None or not os.path.exists(path):
        return {"fixture": None, "reason": "no golden fixture for this arch / step count"}
    ref = torch.from_numpy(np.load(path)['final']).double()[0]
    got = latent0.detach().double().cpu()
    err = float((got - ref).norm() / ref.norm())
    return {"fixture": "tests/golden/%s.npz" % name, "what": "final latent of global sample 0 of the LAST timed step vs the reference's B=1 loop",
            "rel_l2": round(err, 6), "tol": 1e-2, "ok": bool(err < 1e-2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="t23d", choices=["t23d", "i23d", "unet"],
                    help="t23d = BASELINE configs[1] (the metric's config); i23d = configs[2]; unet = the ShapeNet launcher's U-Net denoiser "
                         "(sample_shapenet_car_t23d.sh: ddim250, v-prediction + mixed prediction, no CFG, batch 4) - not a BASELINE config")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (t23d: 8, i23d: 32)")
    ap.add_argument("--sample-steps", type=int, default=None, help="t23d: 250 EulerEDM steps; i23d: num_steps 50 = 49 Euler steps")
    ap.add_argument("--views", type=int, default=None, help="cameras per sample: 40 (T23D video, train_util_diffusion.py:289) / 24 (I23D, flow_matching_trainer.py:637)")
    ap.add_argument("--res", type=int, default=None, help="render resolution (the metric: 256^2; unet: the launcher's 128)")
    ap.add_argument("--arch", default=None)
    ap.add_argument("--dec-arch", default="DiT2-L/2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--ode-method", default="euler", choices=["euler", "heun", "dopri5"],
                    help="i23d workload: euler = configs[2] (50 fixed steps); dopri5 = the released sampler's default (torchdiffeq "
                         "semantics, atol 1e-6, rtol 1e-3: the number of network evaluations is decided by the solver and reported)")
    ap.add_argument("--unfolded-steps", type=int, default=2,
                    help="extra steps after the timed region with the zero-context fold of the unconditional CFG half disabled "
                         "(prints value_unfolded; 0 = skip)")
    ap.add_argument("--dist", action="store_true",
                    help="go through torch.distributed.run + an RCCL process group even with --gpus 1 (the N-GPU code path on one GPU)")
    args = ap.parse_args()
    i23d = args.workload == "i23d"
    unet = args.workload == "unet"
    args.batch = args.batch or (4 if unet else 32 if i23d else 8)
    args.sample_steps = args.sample_steps or (50 if i23d else 250)
    args.views = args.views or (24 if (i23d or unet) else 40)
    args.res = args.res or (128 if unet else 256)
    args.arch = args.arch or ("UNet-ShapeNet(320ch, attn 4,2,1)" if unet else "DiT-PixArt-L/2" if i23d else "DiT-L/2")
    if unet and args.dec_arch == "DiT2-L/2":
        args.dec_arch = "DiT2-B/2"                   # the second entry point's default decoder

    if (args.gpus > 1 or args.dist) and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    from ln3diff_amd import parallel
    from ln3diff_amd.pipeline import T23DPipeline, FlowMatchingEngine, render_pairs
    from ln3diff_amd.synth import orbit_cameras
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:                       # before any rendezvous: a mismatched launch must fail, not hang
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s); refusing to print a line whose n_gpus is not "
                         "what ran" % (args.gpus, env_world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    rank, local_rank, world = parallel.setup_dist(timeout_s=parallel.BENCH_PG_TIMEOUT_S)   # no rank-0-only phase here: fail fast
    assert world == args.gpus
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    # rank 0 creates the weights, every rank receives them by ONE flat RCCL broadcast per dtype
    if unet:
        from ln3diff_amd.guided_diffusion.unet import create_unet
        from ln3diff_amd.synth import fill_module_random_
        _, dec = build_models(dev, "DiT-B/2", args.dec_arch, fill=(rank == 0))
        # DDPM_MODEL_FLAGS / DIFFUSION_FLAGS of shell_scripts/final_release/inference/sample_shapenet_car_t23d.sh (826.7 M parameters)
        dit = create_unet(32, 320, 2, channel_mult='', learn_sigma=False, attention_resolutions='4,2,1', num_heads=8, num_head_channels=-1,
                          num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0, denoise_in_channels=12, denoise_out_channels=12,
                          mixed_prediction=True, use_spatial_transformer=True, transformer_depth=1, context_dim=768, mixing_logit_init=-6.0,
                          roll_out=False).to(dev)
        if rank == 0:
            fill_module_random_(dit, 0, dev)
    elif i23d:
        _, dec = build_models(dev, "DiT-B/2", args.dec_arch, fill=(rank == 0))
        dit = build_i23d(dev, args.arch, fill=(rank == 0), golden_weights=True)
    else:
        dit, dec = build_models(dev, args.arch, args.dec_arch, fill=(rank == 0), golden_weights=True)
    # every collective of the run is exercised before the timed region: an all_reduce of ones (how many ranks RCCL reaches),
    # the flat weight broadcast (timed separately: start-up, not part of a step)
    seen = parallel.ranks_seen()
    if seen != world:
        raise SystemExit("bench.py: all_reduce of ones returned %d on a %d-rank launch" % (seen, world))
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    parallel.broadcast_flat([p.data for p in dit.parameters()] + [p.data for p in dec.parameters()] +
                            [b for b in dec.buffers()], src=0)
    torch.cuda.synchronize()
    bcast_ms = parallel.max_over_ranks((time.perf_counter() - t_b) * 1e3)
    B, Bt = args.batch, args.batch * world
    g = torch.Generator(device=dev).manual_seed(42 if i23d else 41)            # global seed, full batch, then sliced per rank
    z_all = torch.randn(Bt, 12, 32, 32, device=dev, generator=g)
    from ln3diff_amd.synth import synth_input
    gseed = 42 if i23d else 41                                                  # global sample 0 = the golden's sample (inputs from synth_input)
    z_all[0] = synth_input('z', (1, 12, 32, 32), gseed)[0].to(dev)
    lo, hi = parallel.shard_range(Bt, rank, world)
    cams = orbit_cameras(args.views).to(dev)
    if unet:
        from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
        from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
        from ln3diff_amd.pipeline import GuidedDiffusionEngine
        diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, 'ddim%d' % args.sample_steps), betas=gd.get_named_beta_schedule('linear', 1000),
                               model_mean_type=gd.ModelMeanType.V)
        eng = GuidedDiffusionEngine(dit, dec, diff, triplane_scaling_divider=1.0, img_size=args.res, diffusion_input_size=32)
        c_all = torch.randn(Bt, 77, 768, device=dev, generator=g)
        cond = {'crossattn': c_all[lo:hi].contiguous()}

        def sample_fn(lo_, hi_):
            if hi_ <= lo_:
                return torch.empty(0, 12, 32, 32, device=dev)
            return eng.sample(cond, batch_size=hi_ - lo_, use_ddim=True, noise=z_all[lo_:hi_].clone(), clip_denoised=False, unconditional_guidance_scale=1.0)
    elif i23d:
        eng = FlowMatchingEngine(dit, dec, sampling_method=args.ode_method)     # configs[2]: euler, 50 fixed steps
        c_all = {'crossattn': torch.randn(Bt, 256, 2048, device=dev, generator=g), 'vector': torch.randn(Bt, 768, device=dev, generator=g)}
        c_all['crossattn'][0] = synth_input('ca', (1, 256, 2048), gseed)[0].to(dev)
        c_all['vector'][0] = synth_input('v', (1, 768), gseed)[0].to(dev)
        cond = {k: v[lo:hi].contiguous() for k, v in c_all.items()}

        def sample_fn(lo_, hi_):
            if hi_ <= lo_:
                return torch.empty(0, 12, 32, 32, device=dev)
            return eng.sample(cond, None, batch_size=hi_ - lo_, cfg_scale=4.0, num_steps=args.sample_steps, zs=z_all[lo_:hi_].clone())
    else:
        c_all = torch.randn(Bt, 77, 768, device=dev, generator=g)
        c_all[0] = synth_input('c', (1, 77, 768), gseed)[0].to(dev)
        cond = {'crossattn': c_all[lo:hi].contiguous()}
        uc = {'crossattn': torch.zeros_like(cond['crossattn'])}
        pipe = T23DPipeline(dit, dec, num_steps=args.sample_steps, cfg_scale=6.5)

        eng = pipe

        def sample_fn(lo_, hi_):
            if hi_ <= lo_:
                return torch.empty(0, 12, 32, 32, device=dev)
            return pipe.sample_latent(z_all[lo_:hi_].clone(), cond, uc)

    # One step = parallel.sharded_step: this rank's samples are denoised, ONE all_gather of the latents, then this rank's share of
    # the B x V (sample, view) pairs is decoded + rendered (with B >= ranks: the views of its own samples).
    def one_step():
        lat_all, frames, _ = parallel.sharded_step(
            sample_fn, lambda la, pairs: render_pairs(la, eng.rec_model, cams, pairs, eng.triplane_scaling_divider, resolution=args.res),
            Bt, args.views, rank, world)
        return lat_all, frames

    for _ in range(args.warmup):
        out = one_step()
    # dominant-kernel timing INSIDE the timed region: HIP events on the launch stream around the MLP fc1 GEMM of the middle
    # layer, every denoise step (an event pair costs ~1 us of stream time per 12 ms step)
    if unet:
        args.no_probes, args.unfolded_steps, args.no_cpu_baseline = True, 0, True     # the DiT probes / CFG fold / oracle baseline do not apply
    if rank == 0 and not args.no_probes:
        dit._fc1_probe = {'layer': dit.depth // 2, 'events': [], 'max': 4096}
    clk = ClockSampler() if (rank == 0 and not args.no_probes) else None
    if clk is not None:
        clk.__enter__()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tw0 = time.time()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tw1 = time.time()
    dt = parallel.max_over_ranks(dt)
    ok = bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[1]['image_raw']).all())
    out_timed = out
    fc1_events = dit._fc1_probe['events'] if getattr(dit, '_fc1_probe', None) else []      # the timed region's launches only
    dit._fc1_probe = None
    # The same step with the unconditional half's cross-attention NOT folded (what a non-zero uc context costs): a few extra steps
    # after the timed region, timed the same way, so the driver-run record holds both figures.
    value_unfolded = None
    if args.unfolded_steps > 0 and not os.environ.get("LN3D_NO_UC_FOLD"):
        os.environ["LN3D_NO_UC_FOLD"] = "1"
        one_step()                                           # warm-up of the unfolded shapes (workspaces, tile choice)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.unfolded_steps):
            one_step()
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        dtu = parallel.max_over_ranks(time.perf_counter() - t1)
        del os.environ["LN3D_NO_UC_FOLD"]
        value_unfolded = round(Bt * args.unfolded_steps / dtu, 5)
    out = out_timed

    if rank == 0:
        if unet:
            wl = ("ShapeNet launcher (sample_shapenet_car_t23d.sh), NOT a BASELINE config: U-Net denoiser (320 channels, attention at 32^2 / 16^2 / 8^2, "
                  "spatial transformer depth 1, 826.7 M parameters), DDIM %d steps, v-prediction + mixed prediction, no CFG, batch %d per GPU, VAE decode "
                  "%s + conv decoder (the Objaverse decoder class), %d views @ %d^2" % (args.sample_steps, B, args.dec_arch, args.views, args.res))
            metric = "3D samples/sec (ShapeNet U-Net denoiser, ddim250 + triplane decode + 128^2 render)"
        elif i23d:
            evals = ("= %d network evaluations per sample" % (args.sample_steps - 1) if args.ode_method == "euler" else
                     "= %d per sample" % (2 * (args.sample_steps - 1)) if args.ode_method == "heun" else "network evaluations decided by the solver: see 'ode'")
            wl = ("BASELINE configs[2]: %s image-cond I23D, flow-matching ODE %s num_steps %d (%s, each on the CFG-doubled batch), CFG 4.0, "
                  "batch %d per GPU, VAE decode %s + conv decoder, %d views @ %d^2 (64+64 samples/ray)"
                  % (args.arch, args.ode_method, args.sample_steps, evals, B, args.dec_arch, args.views, args.res))
            metric = "3D samples/sec (50-step flow-matching DiT-PixArt-L/2 + triplane decode + 256^2 render)"
        else:
            wl = ("BASELINE configs[1]: %s text-cond T23D, EulerEDM/LegacyDDPM-sigma %d steps, CFG 6.5 (network batch 2B), batch %d "
                  "per GPU, VAE decode %s + conv decoder, %d views @ %d^2 (64+64 samples/ray)"
                  % (args.arch, args.sample_steps, B, args.dec_arch, args.views, args.res))
            metric = "3D samples/sec (250-step DiT-L/2 + 256^2 triplane render)"
        ref_cfg = dict(workload="t23d|i23d") if unet else \
            dict(arch="DiT-PixArt-L/2", sample_steps=50, batch=32, views=24, res=256, ode_method="euler") if i23d else \
            dict(arch="DiT-L/2", sample_steps=250, batch=8, views=40, res=256)
        dev_from = {k: getattr(args, k) for k, v in ref_cfg.items() if getattr(args, k) != v}
        if dev_from:           # a reduced / altered run must not pass for the headline configuration
            metric += " [NOT the baseline configuration: %s]" % ", ".join("%s=%s" % kv for kv in sorted(dev_from.items()))
        rec = {
            "metric": metric, "value": round(Bt * args.steps / dt, 5),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl, "global_batch": Bt, "views": args.views, "res": args.res,
                       "parallelism": "dp%d (independent samples per rank, no in-loop collective)" % world,
                       # the released samplers' unconditional context is all zeros: that half's cross-attention is a per-layer constant
                       # and is folded (exact algebra, tests/test_dit_gpu.py / test_i23d_gpu.py); LN3D_NO_UC_FOLD=1 gives the unfolded figure
                       "cfg_uncond": ("zero unconditional context (released configuration), " +
                                      ("NOT folded (LN3D_NO_UC_FOLD=1)" if os.environ.get("LN3D_NO_UC_FOLD") else
                                       "cross-attention of the unconditional half folded; value_unfolded = the same step without the fold"))},
            "value_unfolded": value_unfolded,
            "finite": ok,
            "ranks_seen": seen, "collectives": parallel.collective_info(), "bcast_ms": round(bcast_ms, 2),
            "golden_check": golden_check(out[0][0], i23d, args.arch, args.sample_steps if (not i23d or args.ode_method == "euler") else -1),
        }
        if not i23d and rec["golden_check"].get("fixture") == "tests/golden/full_edm_ditl2_250.npz":
            pc = picture_check(out[0][0], eng.rec_model, args.dec_arch)
            if pc:
                rec["golden_check"].update(pc)
                rec["golden_check"]["ok"] = bool(rec["golden_check"]["ok"] and pc["rgb_rel_l2"] < 1e-2)
        if i23d:
            rec["ode"] = {"method": args.ode_method}
            st = getattr(eng, "last_ode_stats", None)
            if st:              # adaptive solver (rank 0's shard of the LAST timed step): the solver, not --sample-steps, decides the work
                rec["ode"].update({"nfe": st["nfe"], "steps_attempted": st["steps"], "steps_accepted": st["accepted"], "t_end": round(st["t_end"], 6),
                                   "atol": 1e-6, "rtol": 1e-3, "output_grid": "linspace(0, 1, %d)[-1] by 4th-order dense output" % args.sample_steps})
        if not args.no_probes:
            D = dit.embed_dim
            ev = fc1_events
            rec["clock_timed_region"] = clk.summary(tw0, tw1)        # the denoise loop's own clock / power (the whole step above 600 W)
            tp0 = time.time()
            rec["roofline"] = roofline_probe(dev, 2 * B, D, tokens=768)
            tp1 = time.time()
            if ev:
                us = sum(a.elapsed_time(b) for a, b in ev) / len(ev) * 1e3
                r = rec["roofline"]
                r["isolated_loop_avg_us"] = r["avg_us"]          # >= 0.6 s of back-to-back launches of the same GEMM after the run
                ci = clk.summary(tp0, tp1)
                if ci:
                    r["isolated_loop_sclk_mhz"], r["isolated_loop_power_w"] = ci["sclk_mhz"], ci["power_w"]
                r["avg_us"] = round(us, 2)                       # in situ: mean over the timed region's launches (HIP events)
                r["launches_timed"] = len(ev)
                r["achieved"] = round(r["algorithmic_flop_per_launch"] / (us * 1e-6) / 1e12, 1)
                r["frac"] = round(r["achieved"] / r["peak"], 4)
                cs = rec["clock_timed_region"]
                if cs:          # in situ: the clock of the timed region itself
                    r["sclk_mhz"], r["power_w"], r["power_cap_w"] = cs["sclk_mhz"], cs["power_w"], cs["power_cap_w"]
                    r["frac_at_clock"] = round(r["achieved"] / (r["peak"] * cs["sclk_mhz"] / 2400.0), 4)
            else:
                _with_clock(rec["roofline"], clk, tp0, tp1)
            tp0 = time.time()
            rec["roofline_attention"] = attention_probe(dev, 2 * B, dit.num_heads, 1024 if i23d else 768, D // dit.num_heads, Nq=768)
            _with_clock(rec["roofline_attention"], clk, tp0, time.time())
            rec["roofline_raymarch"] = render_probe(dev, dec)
            tp0 = time.time()
            rec["roofline_gate_residual"] = gate_res_probe(dev, 2 * B, D, tokens=768)
            _with_clock(rec["roofline_gate_residual"], clk, tp0, time.time())
            tp0 = time.time()
            rec["roofline_out_proj"] = out_proj_probe(dev, 2 * B, D, tokens=768)
            _with_clock(rec["roofline_out_proj"], clk, tp0, time.time())
            mp = mfma_probe(dev)
            cs = clk.summary(mp.pop("t0"), mp.pop("t1"))
            if cs:
                mp.update(sclk_mhz=cs["sclk_mhz"], power_w=cs["power_w"])
            rec["mfma_sustained_probe"] = mp
            for key in ("roofline", "roofline_attention", "roofline_gate_residual"):
                r = rec[key]
                r["peak_datasheet"] = r["peak"]
                r["peak_sustained_probe"] = mp["tflops"]
                r["frac_of_sustained"] = round(r["achieved"] / mp["tflops"], 4)
            clk.__exit__()
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(args.arch, args.sample_steps, args.views, args.res, B, i23d=i23d)
        print(json.dumps(rec), flush=True)
    parallel.barrier()
    parallel.shutdown()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:          # a failing rank must take the job down (non-zero exit -> the launcher kills the others), never leave them waiting in a collective
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(e.code if isinstance(e, SystemExit) and isinstance(e.code, int) else 1)
