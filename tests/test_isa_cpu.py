"""ISA guard of the built library (no GPU needed): the code objects inside libln3d_hip.so must not contain a packed-fp32 instruction whose
source is taken from the HIGH half of a register pair through op_sel - the form that made the ray-marcher irreproducible from launch to launch
on gfx950 (profiles/r6_render_opsel.md; tools/check_isa.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_guard_pattern_matches_the_bad_form_only():
    import check_isa
    bad = "v_pk_fma_f32 v[12:13], v[16:17], v[54:55], v[12:13] op_sel:[0,1,0]// 000000006B54: D3B0500C 1C326D10"
    ok = ["v_pk_fma_f32 v[30:31], v[40:41], v[84:85], v[30:31] op_sel_hi:[1,0,1]",
          "v_pk_mul_f32 v[28:29], v[28:29], v[34:35] op_sel_hi:[1,0]",
          "v_pk_add_f32 v[28:29], v[28:29], v[42:43]",
          "v_pk_fma_f16 v1, v2, v3, v4 op_sel:[0,1,0]"]          # 16-bit packed math is not the subject
    assert check_isa.BAD.search(bad)
    assert check_isa.BAD.search("v_pk_mul_f32 v[4:5], v[4:5], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]")
    for line in ok:
        assert not check_isa.BAD.search(line), line


def test_built_library_has_no_high_half_op_sel_on_packed_fp32(hip_lib_path):
    import check_isa
    if not os.path.exists(os.path.join(check_isa.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    n, bad = check_isa.check(hip_lib_path)
    assert n > 1000, "the disassembly found the library's packed instructions"
    assert not bad, bad[:5]
