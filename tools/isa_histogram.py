#!/usr/bin/env python
"""
Instruction histogram of a kernel's time loop from the compiler's assembly (hipcc --save-temps: *-gfx950.s), by the issue
classes measured in profiles/r03_valu_rates.txt (tools/ubench/valu_rates.hip):

    fp64      4 cycles per wave64 instruction per SIMD: v_fma/mul/add/max/ldexp/cvt/..._f64, v_mad_u64_u32, 64-bit shifts
    quarter  16 cycles: v_rcp_f64, v_rsq_f64, v_sqrt_f64
    int32_3op 4 cycles: 32-bit VOP3-only integer ops (v_add3, v_lshl_add, v_bfe, v_bfi, v_perm, v_alignbit, v_mul_*, ...)
    int32     2 cycles back to back (3.5 - 3.9 measured when mixed 1:3 .. 1:1 into an fp64 stream): VOP2 / VOP1 32-bit ops,
              v_bitop3_b32, v_fma_f32, v_cndmask_b32
    trans32   8 cycles: v_exp/log/rcp/rsq/sqrt/sin/cos_f32

    python tools/isa_histogram.py <file.s> <kernel-name substring> [--json]

The time loop is taken to be the innermost loop (a basic block, or run of blocks, ending in a backward branch) with the
most VALU instructions.  Prints the per-iteration histogram, the VALU count, the LDS / SALU / branch counts and the issue
cycles per iteration under (a) the measured per-class costs and (b) every instruction at 4 cycles.
"""
from __future__ import annotations

import json
import re
import sys
from collections import Counter

QUARTER = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_fixup_f64", "v_div_fmas_f64", "v_div_scale_f64"}
TRANS32 = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
INT32_FAST = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_lshlrev_b32",
              "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32",
              "v_cndmask_b32", "v_not_b32", "v_bfrev_b32", "v_min_u32", "v_min_i32", "v_max_i32", "v_accvgpr_write_b32",
              "v_accvgpr_read_b32", "v_add_i32", "v_sub_i32"}
CYCLES = {"fp64": 4.0, "quarter": 16.0, "int32_3op": 4.0, "int32": 2.0, "trans32": 8.0}


def classify(op: str) -> str:
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if base in QUARTER:
        return "quarter"
    if base in TRANS32:
        return "trans32"
    if base in INT32_FAST:
        return "int32"
    if base.endswith("_f64") or "_f64_" in base or base.endswith("_b64") or base.endswith("_u64") or base.endswith("_i64") \
            or base in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return "fp64"
    return "int32_3op"


def function_body(text: str, name: str):
    labels = [m for m in re.finditer(r"^(\S*" + re.escape(name) + r"\S*):\s*(?:;.*)?$", text, re.M)
              if not m.group(1).startswith(".")]
    if not labels:
        raise SystemExit(f"no function matching '{name}'")
    m = labels[0]
    end = text.index("s_endpgm", m.end())
    return m.group(1), text[m.end():end]


def loops(body: str):
    """-> list of (first_label, [instruction lines]) for every backward branch, innermost first"""
    lines = body.split("\n")
    label_at = {}
    instrs = []                       # (index, opcode, full line)
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("//"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label_at[m.group(1)] = len(instrs)
            continue
        if s.startswith("."):
            continue
        instrs.append(s)
    out = []
    for i, s in enumerate(instrs):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", s) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            out.append((m.group(1), instrs[label_at[m.group(1)]:i + 1]))
    return out


def histogram(instrs):
    ops = Counter()
    for s in instrs:
        ops[s.split()[0]] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_readlane") and not k.startswith("v_readfirstlane")
            or k.startswith("v_readlane") or k.startswith("v_readfirstlane")}
    classes = Counter()
    for k, v in valu.items():
        classes[classify(k)] += v
    return ops, valu, classes


def analyse(text: str, kernel: str) -> dict:
    """the histogram of `kernel`'s time loop in the assembly `text`"""
    fn, body = function_body(text, kernel)
    cands = loops(body)
    if not cands:
        raise SystemExit("no loop found")
    # innermost = fewest instructions among those holding the most VALU per instruction; pick the loop with most VALU
    # that contains no other backward branch
    best = None
    for lab, ins in cands:
        inner = sum(1 for s in ins[:-1] if re.match(r"^s_c?branch", s) and any(s.endswith(l) for l, _ in cands if l != lab))
        n_valu = sum(1 for s in ins if s.startswith("v_"))
        if best is None or n_valu > best[2]:
            if not any(lab2 != lab and set(ins2) <= set(ins) and len(ins2) < len(ins) and sum(1 for s in ins2 if s.startswith("v_")) > 20
                       for lab2, ins2 in cands):
                best = (lab, ins, n_valu)
    lab, ins, n_valu = best
    ops, valu, classes = histogram(ins)
    res = {"function": fn, "loop": lab, "instructions": len(ins), "valu": n_valu,
           "lds": sum(v for k, v in ops.items() if k.startswith("ds_")),
           "salu": sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith("s_cbranch") and not k.startswith("s_branch")
                       and not k.startswith("s_waitcnt") and not k.startswith("s_nop")),
           "vmem": sum(v for k, v in ops.items() if k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_") or k.startswith("flat_")),
           "classes": dict(classes),
           "cycles_per_class_model": sum(CYCLES[c] * n for c, n in classes.items()),
           "cycles_all_at_4": 4.0 * n_valu + 12.0 * classes.get("quarter", 0),
           "valu_opcodes": dict(sorted(valu.items(), key=lambda kv: -kv[1])),
           "class_cycles": CYCLES}
    return res


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    res = analyse(open(sys.argv[1]).read(), sys.argv[2])
    fn, lab, ins, n_valu, classes = res["function"], res["loop"], [None] * res["instructions"], res["valu"], res["classes"]
    if "--json" in sys.argv:
        print(json.dumps(res))
        return
    print(f"{fn}\n  loop {lab}: {len(ins)} instructions, {n_valu} VALU, {res['lds']} LDS, {res['salu']} SALU, {res['vmem']} VMEM")
    print("  classes:", dict(classes))
    print(f"  issue cycles per iteration: per-class model {res['cycles_per_class_model']:.0f}, every VALU at 4 (+12 per quarter-rate) "
          f"{res['cycles_all_at_4']:.0f}")
    for k, v in res["valu_opcodes"].items():
        print(f"    {v:4d}  {k:28s} {classify(k)}")


if __name__ == "__main__":
    main()
