#!/usr/bin/env python
"""r6: `bench.py --workload unet` with the U-Net's self-attention forced back onto ln3d_attention_small (the r5 path), for a same-box A/B
against the MFMA route: `python tools/unet_bench_ab.py <bench flags>`."""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ln3diff_amd.guided_diffusion import unet  # noqa: E402
unet._MFMA_MIN_TOKENS = 1 << 30
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
