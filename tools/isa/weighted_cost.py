#!/usr/bin/env python3
"""tools/isa/weighted_cost.py <file.s> <kernel-name-substring> [--blocks] -- static VALU issue cost of a gfx950 kernel.

Every VALU instruction of the kernel's assembly is priced with the issue costs MEASURED by tools/ubench/inst_rate.hip on an
MI355X (profiles/r05_inst_rate.txt), in SIMD cycles per wave64 instruction:
  2.2  v_add/sub (u32, f32), v_mul_f32, v_fma/fmac_f32, v_and/or/xor/not, v_mov, v_ashrrev_i32 -- with VGPR / constant operands
  4.1  everything else: any instruction with an SGPR operand, min/max/med3, compares, v_cndmask, shifts left, bit-field
       ops, 24-bit and 32-bit multiplies, 64-bit integer and double-precision arithmetic, conversions, packed fp32, DPP, SDWA
  8.1  v_rcp_f32 / v_rsq / v_sqrt      16.1  v_rcp_f64
Regions are delimited by `; GRPHASE <name>` comments (asm volatile markers of a scratch copy of the source); --blocks
prints every basic block instead.  Static: a loop body counts once."""
import re
import sys

FAST = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|and_b32|or_b32|xor_b32|not_b32|mov_b32|ashrrev_i32)(_e32|_e64)?$")


def cost(op, operands):
    if not op.startswith("v_"):
        return 0.0
    if re.match(r"v_(rcp|rsq|sqrt)_(iflag_)?f32", op):
        return 8.1
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return 16.1
    if "dpp" in op or "sdwa" in op or " row_" in operands or "sel:" in operands:
        return 4.1
    if FAST.match(op):
        # an SGPR source (s12, s[4:5], vcc, exec) doubles the cost
        srcs = operands.split(",")[1:]
        if any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp)", s) for s in srcs):
            return 4.1
        return 2.2
    return 4.1


def main():
    lines = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    by_block = "--blocks" in sys.argv
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur, order, acc = "entry", [], {}
    for l in lines[start:end]:
        m = re.search(r"; GRPHASE (\w+)", l)
        if m and not by_block:
            cur = "after " + m.group(1)
            continue
        if by_block and re.match(r"^\.LBB\d+_\d+:", l):
            cur = l.split(":")[0] + ("  " + l.split(";")[1].strip() if ";" in l else "")
            continue
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op, _, operands = t.partition(" ")
        d = acc.setdefault(cur, {"valu": 0, "cycles": 0.0, "salu": 0, "lds": 0, "vmem": 0})
        if cur not in order:
            order.append(cur)
        if op.startswith("v_"):
            d["valu"] += 1
            d["cycles"] += cost(op, operands)
        elif op.startswith("ds_"):
            d["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            d["vmem"] += 1
        else:
            d["salu"] += 1
    tot = {"valu": 0, "cycles": 0.0}
    for k in order:
        d = acc[k]
        tot["valu"] += d["valu"]; tot["cycles"] += d["cycles"]
        print(f"{k[:70]:70s} VALU {d['valu']:5d}  cycles {d['cycles']:8.1f}  SALU {d['salu']:4d}  LDS {d['lds']:3d}  VMEM {d['vmem']:3d}")
    print(f"{'total':70s} VALU {tot['valu']:5d}  cycles {tot['cycles']:8.1f}")


if __name__ == "__main__":
    main()
