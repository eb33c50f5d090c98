#!/usr/bin/env python
"""Same-box A/B of whole-pipeline builds: `LN3D_LIB=build/libln3d_x.so python tools/bench_with_lib.py <bench.py flags>` runs bench.py
against an alternative build of the kernel library (box-to-box spread of the headline is +-3 %, larger than most kernel deltas)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ln3diff_amd import _lib  # noqa: E402

if os.environ.get('LN3D_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LN3D_LIB'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
