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


# ---- barriers behind a loop's back edge: an LDS write of the previous iteration must be complete before the barrier ----
spec2 = importlib.util.spec_from_file_location('check_barrier_waits', os.path.join(ROOT, 'tools', 'check_barrier_waits.py'))
cbw = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(cbw)


def test_barrier_checker_flags_the_pattern_hipcc_emitted():
    """what ROCm 7.2 made of the loop of round 4's persistent kernel (docs/experiments/) before the wait was written out: a
    value is stored to LDS at the end of an iteration and the loop header is a bare s_barrier followed by the other waves'
    ds_read"""
    bad = """
.LBB18_6:
	ds_write_b32 v228, v2
.LBB18_7:
	s_or_b64 exec, exec, s[0:1]
	s_mov_b64 s[0:1], 0
.LBB18_8:
	s_and_b64 vcc, exec, s[0:1]
	s_cbranch_vccnz .LBB18_958
.LBB18_9:
	s_barrier
	ds_read_b32 v2, v228
	s_waitcnt lgkmcnt(0)
.LBB18_958:
	s_endpgm
""".splitlines()
    n, problems = cbw.check(bad)
    assert n == 1 and problems
    good = [ln for ln in bad]
    good.insert(good.index('\ts_barrier'), '\ts_waitcnt lgkmcnt(0)')
    assert not cbw.check(good)[1]
    same_block = ['\tds_write_b32 v1, v2', '\ts_barrier']
    assert cbw.check(same_block)[1]
    assert not cbw.check(['\tds_write_b32 v1, v2', '\ts_waitcnt vmcnt(0) lgkmcnt(0)', '\ts_barrier'])[1]


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_every_kernel_barrier_waits_for_lds_writes():
    assert cbw.main() == 0


# ---- hand-counted operand prefetch: nothing may touch an asm-loaded register before the explicit wait ----
spec3 = importlib.util.spec_from_file_location('check_counted_prefetch', os.path.join(ROOT, 'tools', 'check_counted_prefetch.py'))
ccp = importlib.util.module_from_spec(spec3)
spec3.loader.exec_module(ccp)


def test_counted_prefetch_checker_flags_a_copy_behind_an_asm_load():
    """what hipcc made of a prefetch prologue with a conditional load (round 6, while the prefetch was written): the
    loaded quad is moved to another register right behind the asm, before the data has arrived"""
    bad = """
	;;#ASMSTART
	global_load_dwordx4 v[16:19], v98, s[6:7] offset:0
	;;#ASMEND
	v_mov_b32_e32 v86, v16
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	ds_write_b128 v100, v[86:89]
""".splitlines()
    n, problems = ccp.check(bad)
    assert n == 1 and problems
    good = """
	;;#ASMSTART
	global_load_dwordx4 v[16:19], v98, s[6:7] offset:0
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[20:23], v98, s[6:7] offset:512
	;;#ASMEND
	v_add_u32_e32 v99, 1, v98
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	ds_write_b128 v100, v[16:19]
""".splitlines()
    n, problems = ccp.check(good)
    assert n == 2 and not problems
    # the younger load is still in flight behind vmcnt(1)
    early = good[:-1] + ['\tds_write_b128 v100, v[20:23]']
    assert ccp.check(early)[1]


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_no_kernel_touches_a_prefetched_register_before_its_wait():
    assert ccp.main() == 0
