#!/usr/bin/env python3
"""Per-phase VALU instruction table of a demod_rows instantiation, from `hipcc -S`.

    python tools/isa_table.py                       # target build: demod_rows<2,2,normalized,false,false,false>
    python tools/isa_table.py --cond 1              # the plain form of the same design (round 2's default)
    python tools/isa_table.py --tin 0 --cond 0      # BASELINE config 2's build
    python tools/isa_table.py --asm /tmp/k.s        # reuse an assembly file (skips the 1-minute compile)

Counts the VALU instructions of the straight-line FULL body (the 4-row segment every wavefront of a
BASELINE launch runs: the function's largest basic block), by mnemonic and by phase, and prices them
with the issue costs measured on the chip (tools/microbench/valu_rate.hip, profiles/r1_microbench_valu_rate2.txt,
shader cycles per wave64 instruction per SIMD at 8 wavefronts per SIMD).  SQ_INSTS_VALU of a dispatch is
(count per segment) x (segments per dispatch = rows / 4).

Phases are recognised from the mnemonics (the kernel has one use for each):
  unpack     v_and 0x7f7f7f7f, v_add_u32 0x1010101, v_bitop3 0x78, v_cvt_f32_ubyteN, v_pk_add_f32 (plain) or
             v_pk_fma_f32 v, v, s (normalised: alpha * byte - k)
  stencil_in v_pk_fma_f32 with an SGPR-pair first operand (taps in scalar registers)
  discrim    the inline-asm v_pk_mul_f32 / v_pk_fma_f32 with op_sel (discriminator_argument)
  atan2      v_sub |a|,|b|, v_ashrrev 31, v_bitop3 0xca, v_rcp, v_mul_legacy, polynomial pk ops, v_and pi masks
  stencil_out v_mul_f32 / v_fmac_f32 with SGPR taps
  address    v_lshl_add_u64, v_add_co, v_mov (address and constant set-up)
"""
from __future__ import annotations

import argparse
import collections
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
         f"-I{ROOT}/demodulator_amd/csrc", "-S", "--cuda-device-only"]

# measured issue cost, shader cycles per wave64 instruction per SIMD (profiles/r1_microbench_valu_rate2.txt)
COST = {
    "v_pk_fma_f32": 2.12, "v_pk_mul_f32": 2.12, "v_pk_add_f32": 2.12, "v_pk_mov_b32": 2.12,
    "v_cvt_f32_ubyte": 1.92, "v_rcp_f32": 3.99, "v_bitop3_b32": 1.19, "v_and_b32": 1.18, "v_ashrrev_i32": 1.18,
    "v_sub_f32": 1.30, "v_add_f32": 1.30, "v_add_u32": 1.35, "v_mul_f32": 1.12, "v_fmac_f32": 1.12, "v_fma_f32": 1.50,
    "v_fmaak_f32": 1.33, "v_mul_legacy_f32": 1.34, "v_mov_b32": 1.12, "v_mov_b64": 2.0, "v_lshl_add_u64": 2.0,
    "v_readfirstlane_b32": 2.60, "v_cndmask_b32": 1.97, "v_perm_b32": 1.95, "v_min_f32": 2.0, "v_max_f32": 2.0,
}


def cost_of(op: str) -> float:
    base = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)
    base = re.sub(r"ubyte\d$", "ubyte", base)
    return COST.get(base, 1.5)


def phase_of(op: str, line: str, in_asm: bool) -> str:
    if op.startswith("v_cvt_f32_ubyte") or "0x7f7f7f7f" in line or "0x1010101" in line or "bitop3:0x78" in line:
        return "unpack"
    if op.startswith("v_pk_add_f32"):
        return "unpack"
    if in_asm and op.startswith("v_pk_"):
        return "discrim"
    if op.startswith("v_pk_fma_f32"):
        # taps live in scalar registers: the complex stencil's first source is an SGPR pair; the normalised form's
        # unpack alpha * byte - k has its scalar (alpha) as the second source; the polynomial's as the third
        if re.search(r"v_pk_fma_f32 v\[\d+:\d+\], s\[", line):
            return "stencil_in"
        if re.search(r"v_pk_fma_f32 v\[\d+:\d+\], v\[\d+:\d+\], s\[", line):
            return "unpack"
        return "atan2"
    if op.startswith("v_pk_mul_f32"):
        return "atan2"
    if op.startswith(("v_rcp_f32", "v_mul_legacy_f32", "v_ashrrev_i32", "v_sub_f32")) or "bitop3:0xca" in line:
        return "atan2"
    if op.startswith("v_and_b32") and ("0x40490fdb" in line or "0xbfc90fdb" in line or "0x3fc90fdb" in line):
        return "atan2"
    if op.startswith(("v_mul_f32", "v_fmac_f32")) and re.search(r"\bs\d+", line):
        return "stencil_out"
    if op.startswith(("v_lshl_add_u64", "v_add_co", "v_addc", "v_mov", "v_lshlrev", "v_add_u32", "v_lshl_or", "v_lshl_add")):
        return "address"
    return "other"


def compile_asm(extra: list[str]) -> str:
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run([HIPCC] + FLAGS + extra + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    return out


def function_lines(asm: str, symbol_re: str) -> tuple[str, list[str]]:
    lines = open(asm).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(symbol_re, l):
            start, name = i, l.split(":")[0]
        elif start is not None and l.startswith(".Lfunc_end"):
            return name, lines[start:i]
    raise SystemExit(f"no function matches {symbol_re}")


def blocks_of(lines: list[str]):
    """[(label, [(op, line, in_asm)])]: basic blocks in layout order (a label or a branch ends a block)."""
    out, cur, label, in_asm = [], [], "entry", False
    for raw in lines[1:]:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.endswith(":") and line.startswith(".LBB"):
                out.append((label, cur))
                cur, label = [], line[:-1]
            continue
        if line.endswith(":"):
            out.append((label, cur))
            cur, label = [], line[:-1]
            continue
        op = line.split()[0]
        cur.append((op, line, in_asm))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            out.append((label, cur))
            cur, label = [], label + "+"
    out.append((label, cur))
    return [b for b in out if b[1]]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--tin", type=int, default=2)
    ap.add_argument("--tout", type=int, default=2)
    ap.add_argument("--cond", type=int, default=2, help="0 shiftOrigin (folded where a complex stage exists), 1 plain, 2 stencil divided by its first tap (the default with -L), 3 normalizeInput")
    ap.add_argument("--s16", action="store_true")
    ap.add_argument("--span", action="store_true")
    ap.add_argument("--asm", help="assembly file to read instead of compiling")
    ap.add_argument("--flags", default="", help="extra compiler flags")
    ap.add_argument("--rows", type=int, default=1 << 21, help="rows per dispatch for the SQ_INSTS_VALU estimate")
    args = ap.parse_args()
    asm = args.asm or compile_asm(args.flags.split())
    b = lambda v: "1" if v else "0"
    sym = (rf"^_ZN5demod12_GLOBAL__N_110demod_rowsILi{args.tin}ELi{args.tout}ELi{args.cond}ELb0ELb{b(args.s16)}ELb{b(args.span)}EEEvNS_9RowLaunchE:")
    name, lines = function_lines(asm, sym)
    blocks = blocks_of(lines)
    full_label, full = max(blocks, key=lambda kv: sum(1 for (op, _, _) in kv[1] if op.startswith("v_")))
    by_op, by_phase, by_phase_cost = collections.Counter(), collections.Counter(), collections.Counter()
    n_valu, units = 0, 0.0
    for op, line, in_asm in full:
        if not op.startswith("v_"):
            continue
        n_valu += 1
        key = re.sub(r"_e(32|64)$", "", op)
        key = re.sub(r"ubyte\d$", "ubyteN", key)
        by_op[key] += 1
        ph = phase_of(op, line, in_asm)
        by_phase[ph] += 1
        c = cost_of(op)
        by_phase_cost[ph] += c
        units += c
    other = collections.Counter(op for (op, _, _) in full if not op.startswith("v_"))
    print(f"function  {name}")
    print(f"FULL body {full_label}: {len(full)} instructions, {n_valu} VALU = {n_valu / 16:.2f} per output "
          f"(a 4-row segment is 16 outputs per lane)")
    print(f"SQ_INSTS_VALU estimate for {args.rows} rows: {n_valu * args.rows / 4:.3e} (+ prologue)")
    print(f"weighted issue cost {units:.0f} cycles per segment per SIMD share = {units / 16:.1f} per output\n")
    print("| instruction | count | per output |")
    print("|---|---|---|")
    for k, v in by_op.most_common():
        print(f"| {k} | {v} | {v / 16:.2f} |")
    print("\n| phase | VALU | per output | weighted cycles | share |")
    print("|---|---|---|---|---|")
    for k, v in by_phase.most_common():
        print(f"| {k} | {v} | {v / 16:.2f} | {by_phase_cost[k]:.0f} | {100 * by_phase_cost[k] / units:.0f} % |")
    print("\nnon-VALU in the body: " + ", ".join(f"{k} {v}" for k, v in other.most_common(12)))
    m = re.search(r"; NumVgprs: (\d+)", "\n".join(open(asm).read().split(name + ":", 1)[1].split("\n")[:6000]))
    if m:
        print(f"NumVgprs {m.group(1)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
