#!/usr/bin/env python3
"""Companion of check_exec_prologue.py: lists, per kernel of a device assembly file, the vector registers that carry spilled SCALAR registers in their
lanes (v_writelane_b32 / v_readlane_b32) and every OTHER instruction that touches such a register. A carrier that is also ordinary data (or that is
itself spilled to scratch under a partial exec mask) loses scalar values; used in round 5 to look at round 3's hanging build (profiles/r05/notes.md).
usage: check_spill_carriers.py file.s [...]    (report only; exit code 0)"""
import re
import sys


def funcs(path):
    cur, body = None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z[\w]+):", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur:
            body.append(line.rstrip("\n"))
    if cur:
        yield cur, body


def regs_in(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def main(argv):
    for path in argv:
        for name, body in funcs(path):
            carriers = set()
            for line in body:
                m = re.match(r"\s*v_writelane_b32 v(\d+),", line) or re.match(r"\s*v_readlane_b32 s\d+, v(\d+),", line)
                if m:
                    carriers.add(int(m.group(1)))
            if not carriers:
                continue
            other = {}
            for line in body:
                s = line.strip()
                if not s or s.startswith((";", ".")) or s.startswith(("v_writelane_b32", "v_readlane_b32")):
                    continue
                s = s.split(";")[0]
                for r in regs_in(s) & carriers:
                    other.setdefault(r, []).append(line.strip())
            print(f"{path}: {name}: carriers {sorted(carriers)}; touched by other instructions: { {r: len(v) for r, v in other.items()} }")
            for r, v in other.items():
                for x in v[:6]:
                    print(f"      v{r}: {x[:140]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
