#!/usr/bin/env python
"""ISA guard of the built library (r6): no packed-fp32 instruction may take a source from the HIGH half of a register pair through op_sel.

Found with the r6 ray-marcher (profiles/r6_render_opsel.md): on gfx950, `v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[..1..]` (the low
result lane reads the odd register of a 64-bit source pair) returns wrong values in lanes 48 - 63 of the wave - the last of the four 16-lane passes -
from time to time when a second wave shares the SIMD; the same code is bit-reproducible with one wave per SIMD, and with the odd register copied
to an even one first (op_sel_hi-only forms) it is reproducible at any occupancy.  hipcc picks the op_sel form whenever the broadcast value happens
to sit in an odd register (the second / fourth element of a 16-byte load result, an SLP tree), so the library is checked after every build:
every gfx950 code object in the .so's .hip_fatbin section is disassembled and the pattern must not occur.
Usage: python tools/check_isa.py [lib.so]   (exit code 1 + the offending lines if it does)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
BAD = re.compile(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[^\]]*1[^\]]*\]")


def code_objects(lib):
    """the device ELF images inside the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "unused.so")])
        d = open(fat, "rb").read()
    out = []
    for m in re.finditer(b"\x7fELF\x02\x01\x01", d):
        o = m.start()
        e_machine = struct.unpack_from("<H", d, o + 18)[0]
        if e_machine != 224:                      # EM_AMDGPU
            continue
        e_shoff, = struct.unpack_from("<Q", d, o + 40)
        e_shentsize, e_shnum = struct.unpack_from("<HH", d, o + 58)
        out.append(d[o:o + e_shoff + e_shentsize * e_shnum])
    return out


def check(lib):
    bad = []
    n = 0
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f.name], capture_output=True, text=True).stdout
        kern = "?"
        for line in dis.splitlines():
            if line.endswith(">:"):
                kern = line.split("<")[-1][:-2]
            if "v_pk_" in line:
                n += 1
                if BAD.search(line):
                    bad.append((kern, line.strip()))
    return n, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ln3diff_amd", "libln3d_hip.so")
    n, bad = check(lib)
    print(f"{lib}: {n} packed instructions, {len(bad)} with a high-half op_sel source")
    for k, l in bad[:40]:
        print("  ", k, "|", l)
    sys.exit(1 if bad else 0)
