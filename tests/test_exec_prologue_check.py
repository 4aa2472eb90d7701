"""tools/check_exec_prologue.py - the build's gate against the code-generation defect behind round 4's wrong counting render and round 3's hang
(profiles/r05/notes.md section 1): vector spill code placed in front of the s_or_b64 exec that re-converges a block's lanes. The fixture below is
the block as hipcc emitted it for pt_render_kernel<2, true, false, 0> (-DPT_ARGS_AGAIN_EVERYWHERE), registers and offsets as found."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_exec_prologue.py")
spec = importlib.util.spec_from_file_location("check_exec_prologue", TOOL)
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

DEFECT = """\
_Z16pt_render_kernelILi2ELb1ELb0ELi0EEv12PtRenderArgs: ; @kernel
; %bb.190:
\ts_and_saveexec_b64 s[0:1], s[28:29]
\ts_cbranch_execz .LBB14_192
; %bb.191:                              ;   in Loop: Header=BB14_181 Depth=3
\ts_waitcnt vmcnt(0)
\tv_lshl_add_u64 v[136:137], v[48:49], 0, 1
\tv_mov_b64_e32 v[52:53], v[136:137]
.LBB14_192:                             ;   in Loop: Header=BB14_181 Depth=3
\ts_mov_b32 s2, s26
\tscratch_store_dwordx2 off, v[118:119], off offset:236 ; 8-byte Folded Spill
\tscratch_store_dword off, v114, off offset:232 ; 4-byte Folded Spill
\ts_or_b64 exec, exec, s[0:1]
\ts_load_dwordx2 s[74:75], s[24:25], 0x20
\tv_mov_b32_e32 v119, 0
"""
# ordinary code: a spill store inside the divergent region (after a real vector instruction), an exec restore in the middle of a block, a reload behind the restore
CLEAN = """\
kernel_b: ; @kernel_b
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\ts_and_saveexec_b64 s[6:7], vcc
\ts_cbranch_execz .LBB0_3
\ts_and_saveexec_b64 s[4:5], vcc
\ts_cbranch_execz .LBB0_1
.LBB0_1:
\tv_add_f64 v[0:1], v[2:3], v[4:5]
\tscratch_store_dwordx2 off, v[0:1], off offset:8 ; 8-byte Folded Spill
\ts_or_b64 exec, exec, s[4:5]
.LBB0_2:
\ts_mov_b32 s2, s26
\ts_or_b64 exec, exec, s[0:1]
\tscratch_load_dwordx2 v[0:1], off, off offset:8 ; 8-byte Folded Reload
\tv_writelane_b32 v167, s2, 3
.LBB0_3:
\tv_writelane_b32 v167, s2, 4
\ts_or_b64 exec, exec, s[6:7]
"""


def test_the_defect_is_found_and_repaired(tmp_path):
    src = tmp_path / "k.s"
    src.write_text(DEFECT)
    found = chk.scan(str(src))
    assert [d[4].split()[0] for d in found] == ["scratch_store_dwordx2", "scratch_store_dword"]
    assert all(d[1].startswith("_Z16pt_render_kernelILi2ELb1ELb0ELi0E") and d[2] == ".LBB14_192" for d in found)
    fixed, log = chk.repair(DEFECT.splitlines(keepends=True))
    assert len(log) == 1 and log[0][2].startswith("s_or_b64 exec, exec, s[0:1]") and len(log[0][3]) == 2
    text = "".join(fixed)
    # ONE instruction moved, nothing else touched: the exec restore now follows the scalar copy and precedes both stores
    block = text[text.index(".LBB14_192:"):].splitlines()[1:5]
    assert [b.split()[0] for b in block] == ["s_mov_b32", "s_or_b64", "scratch_store_dwordx2", "scratch_store_dword"]
    assert sorted(text.splitlines()) == sorted(DEFECT.splitlines())
    out = tmp_path / "k_fixed.s"
    out.write_text(text)
    assert chk.scan(str(out)) == []


# round 3's hang (profiles/r04/c20_hang_rocgdb.txt: the chunk-sum loop's trip count, v58 = v72, far above 8): the live-range-split copy that SAVES the count sits in
# front of the exec restore, so only the region's lanes save theirs. Block .LBB31_423 of pt_render_simple_kernel<6, false, false, 6, false> as round 3's tree compiles.
R03_HANG = """\
_Z23pt_render_simple_kernelILi6ELb0ELb0ELi6ELb0EEv12PtRenderArgs: ; @kernel
; %bb.367:
\tscratch_store_dwordx2 off, v[0:1], off offset:16 ; 8-byte Folded Spill
\ts_and_saveexec_b64 s[4:5], s[88:89]
\ts_cbranch_execz .LBB31_423
; %bb.368:
\tds_write2st64_b64 v78, v[6:7], v[4:5] offset0:8 offset1:12
.LBB31_423:                             ;   in Loop: Header=BB31_7 Depth=1
\tv_writelane_b32 v79, s96, 18
\tv_mov_b32_e32 v72, v58
\ts_nop 0
\tv_writelane_b32 v79, s97, 19
\ts_or_b64 exec, exec, s[4:5]
\ts_load_dword s33, s[78:79], 0x4
"""


def test_the_copy_form_of_the_defect_round_3s_hang(tmp_path):
    src = tmp_path / "h.s"
    src.write_text(R03_HANG)
    found = chk.scan(str(src))
    assert len(found) == 1 and found[0][4] == "v_mov_b32_e32 v72, v58" and found[0][2] == ".LBB31_423"
    fixed, log = chk.repair(R03_HANG.splitlines(keepends=True))
    text = "".join(fixed)
    block = [b.split()[0] for b in text[text.index(".LBB31_423:"):].splitlines()[1:6]]
    assert block == ["v_writelane_b32", "s_or_b64", "v_mov_b32_e32", "s_nop", "v_writelane_b32"]
    out = tmp_path / "h_fixed.s"
    out.write_text(text)
    assert chk.scan(str(out)) == []
    # NOT the defect (pt_cast_kernel<2> of csrc/pt_api.hip as compiled in round 5; a first version of the checker "repaired" it and would have broken it): the label
    # is the region's BODY, entered with s_cbranch_execnz, laid out behind the join with the join's code duplicated at its end - the move in front of that copy of
    # the exec restore is the region's own code and must stay with the region's lanes
    ok = tmp_path / "ok.s"
    ok.write_text("k: ; @k\n.LBB9_327:\n\ts_or_b64 exec, exec, s[50:51]\n\ts_and_saveexec_b64 s[2:3], s[0:1]\n\ts_cbranch_execnz .LBB9_328\n.LBB9_190:\n"
                  "\ts_or_b64 exec, exec, s[2:3]\n\tv_mov_b32_e32 v0, 0\n\ts_branch .LBB9_332\n.LBB9_328:\n\tv_readlane_b32 s4, v146, 54\n\ts_or_b64 s[56:57], s[0:1], s[4:5]\n"
                  "\tv_mov_b32_e32 v56, v10\n\ts_or_b64 exec, exec, s[2:3]\n\tv_mov_b32_e32 v0, 0\n")
    assert chk.scan(str(ok)) == []
    same, log = chk.repair(ok.read_text().splitlines(keepends=True))
    assert log == [] and "".join(same) == ok.read_text()
    # nor a restore with another region's mask, nor allocator code in a fall-through block
    ok.write_text("k: ; @k\n\ts_and_saveexec_b64 s[6:7], vcc\n\ts_cbranch_execz .LBB0_1\n; %bb.2:\n\tv_mov_b32_e32 v72, v58\n\ts_or_b64 exec, exec, s[4:5]\n.LBB0_1:\n\tv_mov_b32_e32 v1, v2\n\ts_or_b64 exec, exec, s[8:9]\n")
    assert chk.scan(str(ok)) == []


def test_ordinary_code_is_left_alone(tmp_path):
    src = tmp_path / "c.s"
    src.write_text(CLEAN)
    assert chk.scan(str(src)) == []
    fixed, log = chk.repair(CLEAN.splitlines(keepends=True))
    assert log == [] and "".join(fixed) == CLEAN


def test_command_line(tmp_path):
    bad, good, out = tmp_path / "bad.s", tmp_path / "good.s", tmp_path / "out.s"
    bad.write_text(DEFECT)
    good.write_text(CLEAN)
    assert subprocess.run([sys.executable, TOOL, str(good)], capture_output=True).returncode == 0
    r = subprocess.run([sys.executable, TOOL, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "DEFECT" in r.stdout and "Folded Spill" in r.stdout
    r = subprocess.run([sys.executable, TOOL, "--fix", str(bad), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "REPAIRED" in r.stdout and "1 block(s) repaired, 0 defect(s) left" in r.stdout
    assert subprocess.run([sys.executable, TOOL, str(out)], capture_output=True).returncode == 0


def test_the_makefile_builds_every_hip_object_through_the_check():
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert "check_exec_prologue.py --fix" in mk and "--offload-device-only -S" in mk and "-fcuda-include-gpubinary" in mk
    # no rule left that compiles a .hip straight to an object
    for line in mk.splitlines():
        if line.startswith("\t$(HIPCC)") and " -c " in line:
            assert "--cuda-host-only" in line, line
