#!/usr/bin/env python3
"""Build gate against ONE code-generation defect of the AMDGPU backend this library has met twice (round 3's hang of the 80-register
kernel, round 4's wrong counting render of pt_render_kernel<2, true, *, 0>; root-caused in round 5, profiles/r05/notes.md section 1):

    (the same defect with a live-range-split COPY instead of a spill store: round 3's hang, see COPY below)

    a basic block that re-converges a divergent region starts with   s_or_b64 exec, exec, s[N:M]        (SI_END_CF)
    the scalar register allocator, which runs first, may put a live-range-split copy (s_mov_b32 sA, sB) IN FRONT of it - harmless -
    the vector register allocator, which runs second, looks for "the first instruction after the block's prologue" to insert a
    spill store / reload, does not take that copy for part of the prologue, and inserts the VECTOR spill code BEFORE the s_or_b64:

        .LBB14_192:
            s_mov_b32 s2, s26
            scratch_store_dwordx2 off, v[118:119], off offset:236 ; 8-byte Folded Spill     <- executed by the lanes of the region only
            s_or_b64 exec, exec, s[0:1]

    the lanes that sat out the divergent region never store their value; the reload (full exec) hands them whatever the slot held.

This script reads device assembly (hipcc --offload-device-only -S) and reports the vector code only a register allocator inserts - spill stores and
reloads ("Folded Spill" / "Folded Reload"), register-to-register moves - that stands between the label an `s_cbranch_execz` skips a divergent region to (the JOIN) and the `s_or_b64 exec, exec, sX` with that branch's saved exec
mask, with nothing but scalar instructions around it.
usage: check_exec_prologue.py file.s [file.s ...]              exit code 1 if any spill is found in front of an exec restore
       check_exec_prologue.py --fix in.s -o out.s             writes the assembly with every such block REPAIRED - the exec-widening instruction
                                                              moved in front of the spill code, which is where the allocator meant it to be: the
                                                              spill code then runs for every lane that enters the block (profiles/r05/notes.md
                                                              section 1: this one move per block turns the wrong render into the right one) - and
                                                              checks the result; exit code 1 if a defect is left. The Makefile builds the render
                                                              kernels through this (device assembly -> repair -> assemble -> embed)."""
import re
import sys

VECTOR = re.compile(r"^\s*(v_|scratch_|global_|flat_|buffer_|ds_|image_|tbuffer_)")
# what re-converges lanes at the START of a block: SI_END_CF (s_or_b64 exec, exec, saved) and SI_ELSE (s_or_saveexec_b64)
WIDEN = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64)")
BLOCK = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:|[A-Za-z_$][\w$.]*:)")
FUNC = re.compile(r"^([A-Za-z_$][\w$.]*):\s*(;.*)?$")
# lane-independent vector instructions (they ignore exec): scalar values kept in / fetched from lanes of a vector register
EXEC_FREE = re.compile(r"^\s*(v_writelane_b32|v_readlane_b32)")
SPILL = ("Folded Spill", "Folded Reload")
# register-to-register vector moves: what a live-range split by the register allocator looks like (round 3's hang: `v_mov_b32_e32 v72, v58` - the chunk's sample
# count saved across a region - in front of the s_or_b64 exec of block .LBB31_423 of pt_render_simple_kernel<6, false, false, 6, false>)
COPY = re.compile(r"^\s*(v_mov_b32_e32\s+v\d+,\s*v\d+\s*(;.*)?$|v_mov_b64_e32\s+v\[\d+:\d+\],\s*v\[\d+:\d+\]\s*(;.*)?$|v_accvgpr_(read|write)_b32\s)")


SAVE = re.compile(r"^\s*(s_and_saveexec_b64|s_or_saveexec_b64|s_andn2_saveexec_b64|s_xor_saveexec_b64)\s+(s\[\d+:\d+\]|vcc),|^\s*s_xor_b64\s+(s\[\d+:\d+\]|vcc),\s*exec,")
SKIP = re.compile(r"^\s*s_cbranch_execz\s+(\.LBB\d+_\d+)")


def joins(lines):
    """label -> the saved-exec registers of the `s_cbranch_execz label` branches that skip a divergent region to it (the register written by the
    s_and_saveexec_b64 / s_xor_b64 .., exec, .. in front of the branch). Only such a label is a JOIN: lanes that sat the region out arrive there with
    their exec bits cleared, and the block's `s_or_b64 exec, exec, <that register>` brings them back. (A label entered with s_cbranch_execnz is the region's
    own body - possibly with the join's code duplicated behind it -, and vector code in front of an exec restore there is the program's.)"""
    out = {}
    saved = None
    for raw in lines:
        line = raw.rstrip("\n")
        if BLOCK.match(line):
            saved = None
            continue
        m = SAVE.match(line)
        if m:
            saved = m.group(2) or m.group(3)
            continue
        m = SKIP.match(line)
        if m and saved:
            out.setdefault(m.group(1), set()).add(saved)
    return out


def restores(line, regs):
    m = re.match(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)", line)
    return bool(m) and m.group(1) in regs


def allocator_code(line):
    """vector instructions only the register allocator puts at the start of a block: spill stores / reloads and live-range-split copies"""
    return any(t in line for t in SPILL) or COPY.match(line) is not None


def scan(path):
    """A block's PROLOGUE ZONE = its instructions up to the first one that is neither scalar, nor lane-independent, nor vector spill code.
    Spill code inside the zone that is followed, still inside the zone, by an exec-widening instruction is the defect: the register
    allocator placed it at "the start of the block" and landed in front of the instruction that brings the other lanes back.
    (A vector instruction of the program itself in front of an exec restore is ordinary code of the divergent region and ends the zone.)"""
    defects = []
    func = "?"
    label = "?"
    zone = False     # inside the prologue zone of a join block
    pending = []     # allocator code seen in the zone so far
    regs = set()     # the saved-exec registers the join's restore may use
    with open(path, errors="replace") as fh:
        lines = fh.readlines()
    join = joins(lines)
    if True:
        for ln, raw in enumerate(lines, 1):
            line = raw.rstrip("\n")
            m = FUNC.match(line)
            if m and not line.startswith(".L"):
                func = m.group(1)
            if BLOCK.match(line):
                label = line.split(":")[0].strip()
                regs = join.get(label, set())
                pending, zone = [], bool(regs)   # (only a JOIN opens a zone: see joins())
                continue
            s = line.strip()
            if not s or s.startswith((";", ".", "//")):
                continue
            if not zone:
                continue
            if WIDEN.match(line):
                if restores(line, regs):
                    for (pl, pt) in pending:
                        defects.append((path, func, label, pl, pt.strip(), s))
                pending, zone = [], False
            elif VECTOR.match(line) and not EXEC_FREE.match(line):
                if allocator_code(line):
                    pending.append((ln, line))
                else:
                    zone = False
            elif re.match(r"^\s*(s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm|s_waitcnt|s_nop)", line):
                if not re.match(r"^\s*(s_waitcnt|s_nop)", line):
                    zone = False
    return defects


def repair(lines):
    """Moves, in every block whose prologue zone holds vector spill code in front of its exec-widening instruction, that instruction in front
    of the first such spill instruction. Returns (new lines, [(function, block, moved instruction, spill instructions)])."""
    out, log = [], []
    func, label = "?", "?"
    zone = False
    first_spill = None   # index in `out` of the first allocator instruction of the current join block's zone
    spills = []
    join = joins(lines)
    regs = set()
    for raw in lines:
        line = raw.rstrip("\n")
        m = FUNC.match(line)
        if m and not line.startswith(".L"):
            func = m.group(1)
        if BLOCK.match(line):
            label = line.split(":")[0].strip()
            regs = join.get(label, set())
            zone, first_spill, spills = bool(regs), None, []
            out.append(raw)
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or not zone:
            out.append(raw)
            continue
        if WIDEN.match(line):
            # (SI_ELSE is two instructions, s_or_saveexec_b64 + s_xor_b64 exec: not moved - such a block stays a reported defect and fails the build)
            if first_spill is not None and restores(line, regs):
                out.insert(first_spill, raw)
                log.append((func, label, s, [t.strip() for t in spills]))
            else:
                out.append(raw)
            zone = False
            continue
        if VECTOR.match(line) and not EXEC_FREE.match(line):
            if allocator_code(line):
                if first_spill is None:
                    first_spill = len(out)
                spills.append(line)
            else:
                zone = False
        elif re.match(r"^\s*(s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm)", line):
            zone = False
        out.append(raw)
    return out, log


def main(argv):
    if argv and argv[0] == "--fix":
        if len(argv) != 4 or argv[2] != "-o":
            print(__doc__)
            return 2
        with open(argv[1], errors="replace") as fh:
            lines = fh.readlines()
        fixed, log = repair(lines)
        with open(argv[3], "w") as fh:
            fh.writelines(fixed)
        for (func, label, moved, spills) in log:
            print(f"REPAIRED {argv[1]}: {func} block {label}: `{moved}` moved in front of {len(spills)} spill instruction(s): {'; '.join(spills)}")
        left = scan(argv[3])
        for (path, func, label, ln, text, restore) in left:
            print(f"DEFECT LEFT {path}:{ln}: {func} block {label}: spill code in front of `{restore}`: {text}")
        print(f"{argv[1]}: {len(log)} block(s) repaired, {len(left)} defect(s) left")
        return 1 if left else 0
    bad = 0
    for p in argv:
        d = scan(p)
        for (path, func, label, ln, text, restore) in d:
            print(f"DEFECT {path}:{ln}: {func} block {label}: spill code in front of `{restore}`: {text}")
        print(f"{p}: {len(d)} vector spill instruction(s) in front of a block's exec restore")
        bad += len(d)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
