#!/usr/bin/env python3
"""Static check of the hand-counted waits in demod_rows (kernels.hip).

The streaming kernel issues its row loads from inline asm, invisible to hipcc's
own s_waitcnt bookkeeping, and waits for them with hand-counted `s_waitcnt
vmcnt(N)` statements.  The hazard is the compiler
reading or overwriting a destination register of such a load before the wait
that covers it (it believes the value is there at ;;#ASMEND).  This script
compiles kernels.hip to gfx950 assembly and replays every demod_rows function
along EVERY path of its control-flow graph (rows of a clipped segment are
skipped by wave-uniform branches): vector-memory operations retire in issue
order, so after `s_waitcnt vmcnt(N)` everything but the N youngest operations
has landed; any instruction that touches a register of a load that has not
landed yet is reported.

    python tools/check_asm_waits.py            # exit code 1 on a violation
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "demodulator_amd", "csrc", "kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm",
         "-amdgpu-sched-strategy=max-ilp", "-Wno-pass-failed", f"-I{ROOT}/include",
         f"-I{ROOT}/demodulator_amd/csrc", "-S", "--cuda-device-only"] + os.environ.get("DEMOD_CHECK_FLAGS", "").split()

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text: str) -> set[int]:
    out: set[int] = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse(lines):
    """[(line number, op, operands, in_asm)] of the function's instructions and {label: index}."""
    insts, labels = [], {}
    in_asm = False
    for ln, raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(insts)
            continue
        if line.startswith("."):
            continue
        op = line.split()[0]
        insts.append((ln, op, line[len(op):], in_asm, line))
    return insts, labels


def check_function(name: str, lines: list) -> tuple:
    """Walk every path through the function's control-flow graph (both outcomes of each
    conditional branch) with the queue of in-flight vector-memory operations as state."""
    insts, labels = parse(lines)
    problems: dict = {}
    n_asm_loads = sum(1 for (_, op, _, in_asm, _) in insts if in_asm and op.startswith("global_load"))
    n_waits = sum(1 for (_, op, _, in_asm, line) in insts if in_asm and op.startswith("s_waitcnt") and "vmcnt" in line)
    seen = set()
    stack = [(0, ())]     # (instruction index, queue: tuple of (kind, frozenset(dest regs)), oldest first)
    steps = 0
    while stack:
        pc, queue = stack.pop()
        queue = list(queue)
        while pc < len(insts):
            key = (pc, tuple(queue))
            if key in seen:
                break
            seen.add(key)
            steps += 1
            if steps > 2_000_000:
                problems[(name, -1)] = f"{name}: path enumeration did not terminate"
                return list(problems.values()), n_asm_loads, n_waits
            ln, op, operands, in_asm, line = insts[pc]
            nxt = pc + 1
            if op.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:
                    del queue[: max(0, len(queue) - int(m.group(1)))]
            elif op == "s_endpgm":
                break
            elif op == "s_branch":
                nxt = labels[operands.strip()]
            elif op.startswith("s_cbranch"):
                stack.append((labels[operands.strip().split(",")[-1].strip()], tuple(queue)))
            elif op.startswith(("s_", "ds_")) and not op.startswith("ds_"):
                pass
            else:
                pending = set().union(*(r for k, r in queue if k == "asmload")) if queue else set()
                if op.startswith(("global_load", "buffer_load", "flat_load")):
                    parts = operands.split(",")
                    dest = regs(parts[0])
                    touched = regs(",".join(parts[1:])) | dest
                    queue.append(("asmload" if in_asm else "load", frozenset(dest)))
                elif op.startswith(("global_store", "buffer_store", "flat_store")):
                    touched = regs(operands)
                    queue.append(("store", frozenset()))
                else:
                    touched = regs(operands)
                if touched & pending:
                    problems[(name, ln)] = (f"{name}:{ln}: `{line}` touches registers of a load still in flight: "
                                            f"{sorted(touched & pending)}")
                if len(queue) > 48:          # loops of the bounds-checked build (no asm loads there)
                    del queue[: len(queue) - 48]
            pc = nxt
    return list(problems.values()), n_asm_loads, n_waits


def main() -> int:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "kernels.s")
        subprocess.run([HIPCC, *FLAGS, SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    funcs: dict[str, list] = {}
    cur = None
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_ZN5demod\S*demod_rows\S*):", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                cur = None
                continue
            funcs[cur].append((i, l))
    total_loads = total_waits = 0
    problems = []
    for name, lines in funcs.items():
        p, nl, nw = check_function(name, lines)
        problems += p
        total_loads += nl
        total_waits += nw
    print(f"checked {len(funcs)} demod_rows instantiations: {total_loads} asm loads, {total_waits} hand-counted waits, "
          f"{len(problems)} violations")
    for p in problems[:20]:
        print("  ", p)
    return 1 if problems or not funcs or total_loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
