"""The static wait-state audit of the inline-asm statements (tools/asm_hazard_audit.py): the checker flags the hazard classes it
names on hand-written snippets, and the assembly hipcc emits for the shipped sources has none of them (no GPU needed)."""
import importlib.util
import os
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('asm_hazard_audit', os.path.join(ROOT, 'tools', 'asm_hazard_audit.py'))
audit_mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(audit_mod)

GOOD = """
	v_readfirstlane_b32 s5, v3
	s_add_i32 s4, s66, s35
	s_add_i32 s5, s5, s76
	;;#ASMSTART
	s_mov_b32 m0, s4
	s_nop 0
	buffer_load_dwordx4 v1, s[72:75], s5 offen lds
	;;#ASMEND
	;;#ASMSTART
	buffer_store_dwordx4 v[10:13], v1, s[76:79], s6 offen
	s_nop 1
	;;#ASMEND
	v_mov_b32 v10, 0
	;;#ASMSTART
	s_nop 1
	v_add_f32_dpp v10, v16, v16 row_ror:8 row_mask:0xf bank_mask:0xf
	;;#ASMEND
"""


LOAD = """
	;;#ASMSTART
	global_load_dwordx4 v[6:9], v[2:3], off
	;;#ASMEND
	v_add_u32_e32 v1, 1, v1
%s
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f32_e32 v10, v6, v7
"""


def findings(text):
    return audit_mod.audit('snippet.s', text)[0]


def test_checker_passes_the_padded_forms():
    assert findings(GOOD) == []


def test_checker_flags_each_hazard_class():
    # S: the round-5 bug — a 16-byte store whose data registers the next VALU instruction may overwrite
    bad_store = GOOD.replace('offen\n\ts_nop 1\n', 'offen\n')
    assert [f for f in findings(bad_store) if ' S: ' in f]
    # G: the soffset SGPR comes straight from a lane read (e.g. a compiler SGPR-spill reload), 3 wait states ahead of the load
    bad_sgpr = GOOD.replace('\ts_add_i32 s5, s5, s76\n', '')
    assert [f for f in findings(bad_sgpr) if ' G: ' in f]
    # ... five wait states are enough
    assert findings(bad_sgpr.replace('\ts_add_i32 s4, s66, s35\n', '\ts_add_i32 s4, s66, s35\n\ts_nop 2\n')) == []
    # M: the LDS-DMA right behind the write of M0
    assert [f for f in findings(GOOD.replace('\ts_nop 0\n', '')) if ' M: ' in f]
    # D: a DPP read of a register the previous VALU instruction wrote, no wait states
    assert [f for f in findings(GOOD.replace('\ts_nop 1\n\tv_add_f32_dpp', '\tv_add_f32_dpp')) if ' D: ' in f]
    assert [f for f in findings(GOOD.replace('\tv_mov_b32 v10, 0\n', '\tv_cmpx_gt_u32_e32 16, v0\n')) if ' D: ' in f]


def test_checker_follows_an_asm_load_to_its_wait():
    assert findings(LOAD % '') == []
    assert [f for f in findings(LOAD % '\tv_mov_b32_e32 v20, v7') if ' L: ' in f and 'read' in f]            # copied before it landed
    assert [f for f in findings(LOAD % '\tv_mov_b32_e32 v8, 0') if ' L: ' in f and 'overwritten' in f]        # reused before it landed
    # the zero-initialising arm of an `if (bias) load; else zero` diamond is not on the loads' path
    assert findings(LOAD % '\ts_cbranch_vccnz .LBB0_9\n\tv_mov_b32_e32 v8, 0\n.LBB0_9:') == []
    assert [f for f in findings(LOAD.replace('\ts_waitcnt vmcnt(0)', '\ts_nop 0') % '\ts_endpgm') if 'never waited' in f]


@pytest.mark.timeout(900)
def test_shipped_sources_have_no_inline_asm_hazard():
    if not os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    with tempfile.TemporaryDirectory() as tmp:
        files = audit_mod.compile_all(tmp)
        assert len(files) >= 9, files            # every kernel family that issues LDS-DMA or counted waits from asm
        total_vmem = 0
        for f in files:
            found, _n_asm, n_vmem = audit_mod.audit(os.path.basename(f), open(f).read())
            assert found == [], '\n'.join(found)
            total_vmem += n_vmem
        assert total_vmem > 1000                 # the audit looked at the asm blocks, not at an empty stream
