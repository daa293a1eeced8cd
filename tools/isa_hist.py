#!/usr/bin/env python3
"""Instruction-class histogram of one kernel's gfx950 ISA, loop by loop.

  make -C sdirt_amd/csrc        (leaves the device ISA of every translation unit in csrc/obj/)
  python tools/isa_hist.py sdirt_amd/csrc/obj/sdirt_psf-hip-amdgcn-amd-amdhsa-gfx950.s \
         'k_psf_lrILb1ELb0EN5sdirt4LeanELb1' [--json out.json]

Reads the compiler's `.s` (hipcc -S --cuda-device-only), cuts out the kernel whose mangled name
contains the pattern, assigns every instruction to its innermost loop using LLVM's own block
comments ("=>This Loop Header: Depth=N" / "in Loop: Header=BBx_y Depth=N") and counts classes:

  valu        vector ALU, VGPR / inline-constant / literal operands only
  valu_s      vector ALU that reads or writes the scalar file: an SGPR or vcc SOURCE operand,
              v_cmp (writes vcc / an SGPR pair), v_cndmask with an SGPR-pair mask, plus v_med3_*
  cndmask_vcc v_cndmask_b32 with the implicit vcc mask (VOP2)
  valu_trans  v_rcp/v_rsq/v_sqrt/v_sin/v_cos/v_exp/v_log
  valu_f64    fp64 vector ops
  lane        v_readlane / v_writelane / v_readfirstlane (SGPR spill traffic and uniform moves)
  salu, smem (s_load), s_nop, s_waitcnt, branch, vmem, lds

`simd_cyc` prices one pass through the block with the per-SIMD costs MEASURED on gfx950 at 8 waves
per SIMD by tools/form_bench.hip (profiles/r02/form_bench.txt): valu 2.25, valu_s / valu_f64 / lane 4.2,
cndmask_vcc 2.25, valu_trans 8.2, salu / branch / smem / s_waitcnt 2.3, s_nop 1, vmem / lds 2.3 -- an UPPER
estimate of SIMD issue time (in the product the half- and quarter-rate forms hide behind the other
waves: 2.24 cycles per instruction of any kind), not a latency.  The Newton loops are
the innermost loops of the surface loop; their rows are what VERDICT r01 item 3 asks to be tracked.
"""
import argparse
import json
import re
from collections import OrderedDict, defaultdict

TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_(f32|f16|legacy)")
F64 = re.compile(r"^v_\w+_f64|^v_cvt_f64|^v_cvt_\w+_f64")


COST = dict(valu=2.25, valu_s=4.2, cndmask_vcc=2.25, valu_trans=8.2, valu_f64=4.2, lane=4.2, salu=2.3,
            smem=2.3, s_nop=1.0, s_waitcnt=2.3, branch=2.3, vmem=2.3, lds=2.3, barrier=2.3, other=2.3)
SREG = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|exec)\b")


def classify(op, text=""):
    c = classify_op(op)
    if c != "valu":
        return c
    args = text.split(None, 1)[1] if " " in text.strip() else ""
    args = args.split(";")[0]
    if op.startswith("v_cndmask"):
        return "valu_s" if SREG.search(args) else "cndmask_vcc"
    if op.startswith("v_cmp") or op.startswith("v_med3") or op.startswith("v_div_scale") or op.startswith("v_div_fmas"):
        return "valu_s"
    return "valu_s" if SREG.search(args) else "valu"


def classify_op(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane"
    if op.startswith("v_"):
        if TRANS.match(op):
            return "valu_trans"
        if F64.match(op):
            return "valu_f64"
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_endpgm", "s_setpc_b64"):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def kernel_body(lines, pattern):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and pattern in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]):
            if re.match(r"^_Z\S+:", l):
                start = i
                break
    if start is None:
        raise SystemExit(f"no kernel matching {pattern!r}")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def histogram(body):
    """-> OrderedDict loop_key -> {depth, classes...}; loop_key = header label or 'straight'."""
    loops = OrderedDict()
    cur, last_label = ("straight", 0), None
    hdr = re.compile(r"^(\.LBB\d+_\d+):")
    for l in body:
        m = hdr.match(l)
        if m:
            label = last_label = m.group(1)[2:]
            d = re.search(r"Loop Header: Depth=(\d+)", l)
            i = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if d:
                cur = (label, int(d.group(1)))
            elif i:
                cur = (i.group(1), int(i.group(2)))
            elif "Parent Loop" not in l:
                cur = ("straight", 0)
            continue
        t = l.strip()
        if t.startswith(";") and "Loop Header: Depth=" in l and last_label:       # header of a loop nested in others:
            d = re.search(r"Loop Header: Depth=(\d+)", l)          # the label line names the parents
            cur = (last_label, int(d.group(1)))
            continue
        if l.startswith(";") and "in Loop: Header=" in l:          # "; %bb.N:  ; in Loop: Header=..."
            i = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            cur = (i.group(1), int(i.group(2)))
            continue
        if t.startswith("; %bb.") and "in Loop" not in t:
            cur = ("straight", 0)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        rec = loops.setdefault(cur[0], defaultdict(int))
        rec["depth"] = cur[1]
        rec[classify(op, t)] += 1
    return loops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("pattern")
    ap.add_argument("--json")
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    body = kernel_body(open(args.asm).read().splitlines(), args.pattern)
    loops = histogram(body)
    cols = ["valu", "valu_s", "cndmask_vcc", "valu_trans", "valu_f64", "lane", "salu", "smem", "s_nop",
            "s_waitcnt", "branch", "vmem", "lds"]
    total = defaultdict(int)
    print(f"{'loop':<10}{'depth':>6}" + "".join(f"{c[:9]:>10}" for c in cols) + f"{'simd_cyc':>10}")
    out = {"kernel_pattern": args.pattern, "label": args.label, "cost_per_instr": COST, "loops": {}}
    for k, rec in loops.items():
        cyc = round(sum(COST[c] * rec[c] for c in cols), 1)
        print(f"{k:<10}{rec['depth']:>6}" + "".join(f"{rec[c]:>10}" for c in cols) + f"{cyc:>10}")
        out["loops"][k] = dict(depth=rec["depth"], simd_cycles=cyc, **{c: rec[c] for c in cols})
        for c in cols:
            total[c] += rec[c]
    print(f"{'TOTAL':<10}{'':>6}" + "".join(f"{total[c]:>10}" for c in cols))
    out["total_static"] = dict(total)
    for l in body:
        m = re.search(r"; (NumSgprs|NumVgprs|ScratchSize|Occupancy|SGPRBlocks|LDSByteSize): (\d+)", l)
        if m:
            out[m.group(1)] = int(m.group(2))
    spills = sum(1 for l in body if "SGPR spill to VGPR lane" in l)
    out["sgpr_spill_vgprs"] = spills
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
