"""scripts/check_exec_prologue.py -- the build's guard against AGPR copies that the register allocator places at the top of a join
block in front of the `s_or_b64 exec` restoring the enclosing region's lanes -- on a synthetic snippet of the
pattern (the block that faulted on the box, shortened) and on harmless neighbours of it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "scripts", "check_exec_prologue.py")

BAD = """
_ZN4pdmp11lane_kernelILi0ELi10ELb1ELb0ELi2ELi4EEEvNS_11GroupParamsEPKi:
	s_and_saveexec_b64 s[12:13], vcc
	s_cbranch_execz .LBB4_141
; %bb.140:
	global_store_dwordx2 v[4:5], a[26:27], off offset:32
.LBB4_141:                              ;   in Loop: Header=BB4_5 Depth=1
	v_accvgpr_write_b32 a30, v96
	v_accvgpr_write_b32 a28, v82
	v_writelane_b32 v255, s80, 45
	v_accvgpr_write_b32 a31, v97
	v_readlane_b32 s16, v254, 4
	s_or_b64 exec, exec, s[12:13]
	s_ashr_i32 s2, s52, 31
	s_endpgm
"""

GOOD = """
_ZN4pdmp11lane_kernelILi0ELi13ELb1ELb0ELi1ELi4EEEvNS_11GroupParamsEPKi:
	s_and_saveexec_b64 s[12:13], vcc
	s_cbranch_execz .LBB1_10
; %bb.9:
	v_accvgpr_write_b32 a3, v7                  ; a copy INSIDE the region: runs for the lanes it is meant for
	global_store_dwordx2 v[4:5], v[6:7], off
.LBB1_10:
	v_writelane_b32 v255, s80, 45           ; lane operations ignore exec
	v_readlane_b32 s16, v254, 4
	s_or_b64 exec, exec, s[12:13]
	v_accvgpr_write_b32 a30, v96            ; after the restore: every lane
.LBB1_11:                               ; a block that sets its own mask before it copies: the join's prologue ends at the exec write
	v_cmp_lt_i32_e32 vcc, 0, v51
	s_and_saveexec_b64 s[0:1], vcc
	v_accvgpr_read_b32 v2, a22
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
"""

# round 4 (fail closed): a prologue that MIXES the allocator's copies with other vector work is flagged too -- round 3's rule wanted copies
# only and would have let it through; a block reviewed by hand is let through by name
MIXED = """
_ZN4pdmp11lane_kernelILi2ELi13ELb1ELb0ELi1ELi4EEEvNS_11GroupParamsEPKi:
.LBB3_7:
	v_add_f64 v[12:13], v[8:9], -v[14:15]
	v_accvgpr_write_b32 a5, v12
	s_or_b64 exec, exec, s[16:17]
	s_endpgm
"""


def _run(tmp_path, text, name):
    f = tmp_path / name
    f.write_text(text)
    return subprocess.run([sys.executable, CHECK, str(f)], capture_output=True, text=True)


def test_checker_flags_the_miscompiled_join(tmp_path):
    r = _run(tmp_path, BAD, "bad.s")
    assert r.returncode == 1
    assert r.stdout.count("v_accvgpr_write_b32") == 3 and ".LBB4_141" in r.stdout


def test_checker_passes_harmless_neighbours(tmp_path):
    r = _run(tmp_path, GOOD, "good.s")
    assert r.returncode == 0 and r.stdout == ""


def test_checker_fails_closed_on_a_mixed_prologue_and_honours_the_allow_list(tmp_path):
    r = _run(tmp_path, MIXED, "mixed.s")
    assert r.returncode == 1 and ".LBB3_7" in r.stdout
    f = tmp_path / "mixed.s"
    ok = subprocess.run([sys.executable, CHECK, str(f)], capture_output=True, text=True,
                        env=dict(os.environ, PDMP_EXEC_PROLOGUE_ALLOW="lane_kernelILi2ELi13:.LBB3_7"))
    assert ok.returncode == 0 and ok.stdout == ""
