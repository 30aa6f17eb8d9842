"""The hand-written fp64 DPP instructions (diagonal-block pivot chain) sit inside inline asm, where the compiler's hazard
recogniser does not look: tools/check_dpp_hazards.py re-checks the emitted gfx950 assembly (CPU only: hipcc cross-compiles)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('check_dpp_hazards', os.path.join(ROOT, 'tools', 'check_dpp_hazards.py'))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_checker_flags_a_dpp_read_right_after_a_valu_write():
    bad_asm = """
    v_mul_f64 v[2:3], v[8:9], v[10:11]
    v_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
    """
    n, bad = chk.check(bad_asm)
    assert n == 1 and len(bad) == 1
    one_state = """
    v_mul_f64 v[2:3], v[8:9], v[10:11]
    s_nop 0
    v_mov_b64_dpp v[4:5], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf
    """
    assert len(chk.check(one_state)[1]) == 1
    ok_asm = """
    v_mul_f64 v[2:3], v[8:9], v[10:11]
    s_nop 1
    v_mov_b64_dpp v[4:5], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf
    v_fmac_f64_dpp v[12:13], v[2:3], v[4:5] row_newbcast:2 row_mask:0xf bank_mask:0xf
    """
    n, bad = chk.check(ok_asm)
    assert n == 2 and not bad      # src1 of the second one was just written, but only src0 goes through the DPP network


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_product_kernels_have_no_dpp_hazard():
    assert chk.main() == 0
