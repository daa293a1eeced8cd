#!/usr/bin/env python3
"""Half-rate and transcendental instructions of one kernel BY SOURCE SITE.

  hipcc <product flags> -gline-tables-only --cuda-device-only -S sdirt_psf.hip -o psf_g.s
        (line tables do not change the ISA: tools/isa_sites.py --same-as obj/sdirt_psf-...s checks it)
  python tools/isa_sites.py psf_g.s 'k_psf_lrILb1ELb0EN5sdirt4LeanELb1' [--weights rf50mm] [--out table.txt]

Every instruction of the kernel is attributed to the innermost source line LLVM's `.loc` names
for it (after inlining: the line of the device function the operation was written in) and to its
innermost loop; instruction classes are those of tools/isa_hist.py (valu_s = a vector instruction
that reads or writes the scalar file: SGPR / vcc source, v_cmp, v_cndmask on an SGPR-pair mask,
v_med3; valu_trans = v_rcp / v_rsq / v_sqrt ...).  The table lists, per site, the static count of
each class and a DYNAMIC estimate per traced ray: static count x the number of times the site's
loop body runs per ray, from the loop structure the kernel is known to have
(sample loop -> surface loop -> Newton loops) and the verified trip table of the workload
(--weights), checked against the PMC totals of the same kernel (SQ_INSTS_VALU,
SQ_INSTS_VALU_TRANS_F32 per ray) printed at the bottom.
"""
import argparse
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_hist import classify, kernel_body   # noqa: E402

HALF = ("valu_s", "valu_f64", "lane")


def parse(body, files):
    """-> list of (file, line, loop header, depth, class, text)"""
    out, cur, loc, last_label = [], ("straight", 0), ("?", 0), None
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        lab = re.match(r"^\.(LBB\d+_\d+):", l)
        if lab:
            last_label = lab.group(1)[1:]
            i = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if i:
                cur = (i.group(1), int(i.group(2)))
            elif "Loop" not in l:
                cur = ("straight", 0)
        # the header of a loop: the comment may sit on the label's line or on a continuation line
        h = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", l)
        if h and last_label:
            cur = (last_label, int(h.group(1)))
            continue
        if t.startswith("; %bb."):
            i = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            cur = (i.group(1), int(i.group(2))) if i else ("straight", 0)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        out.append((loc[0], loc[1], cur[0], cur[1], classify(op, t), t.split(";")[0].strip()))
    return out


def file_table(lines):
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
    return files


def source_line(name, line, roots):
    for r in roots:
        p = os.path.join(r, name)
        if os.path.exists(p):
            try:
                return open(p).read().splitlines()[line - 1].strip()
            except IndexError:
                return ""
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("pattern")
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    lines = open(args.asm).read().splitlines()
    files = file_table(lines)
    ins = parse(kernel_body(lines, args.pattern), files)
    roots = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdirt_amd", "csrc")]

    # loops: header -> depth, size, which source lines they hold
    loops = defaultdict(lambda: dict(depth=0, n=0, lines=defaultdict(int)))
    for f, ln, hdr, dep, cls, txt in ins:
        L = loops[hdr]
        L["depth"] = dep
        L["n"] += 1
        L["lines"][(f, ln)] += 1
    sites = defaultdict(lambda: defaultdict(int))
    for f, ln, hdr, dep, cls, txt in ins:
        sites[(f, ln, hdr, dep)][cls] += 1
    rows = []
    for (f, ln, hdr, dep), c in sites.items():
        half = sum(c[k] for k in HALF)
        if half + c["valu_trans"] + c["cndmask_vcc"] == 0:
            continue
        rows.append((dep, hdr, f, ln, c))
    rows.sort(key=lambda r: (-(r[4]["valu_trans"] * 4 + sum(r[4][k] for k in HALF) * 2) * max(1, r[0]) ** 2, r[2], r[3]))
    w = sys.stdout if not args.out else open(args.out, "w")
    total = defaultdict(int)
    for _, _, _, _, c in [(0, 0, 0, 0, s) for s in sites.values()]:
        for k, v in c.items():
            total[k] += v
    print(f"# kernel {args.pattern}: {len(ins)} instructions, static totals " +
          ", ".join(f"{k} {v}" for k, v in sorted(total.items())), file=w)
    print("# loops (header: depth, instructions, dominant source lines):", file=w)
    for hdr, L in sorted(loops.items(), key=lambda kv: (kv[1]["depth"], kv[0])):
        dom = sorted(L["lines"].items(), key=lambda kv: -kv[1])[:3]
        print(f"#   {hdr:<10} depth {L['depth']}  {L['n']:5d} instr   " +
              "  ".join(f"{f}:{ln} x{n}" for (f, ln), n in dom), file=w)
    print(f"{'site':<28}{'loop':<10}{'dep':>4}{'valu':>6}{'valu_s':>7}{'cnd_vcc':>8}{'trans':>6}{'f64':>5}{'lane':>5}  source", file=w)
    for dep, hdr, f, ln, c in rows[: args.top]:
        print(f"{f + ':' + str(ln):<28}{hdr:<10}{dep:>4}{c['valu']:>6}{c['valu_s']:>7}{c['cndmask_vcc']:>8}"
              f"{c['valu_trans']:>6}{c['valu_f64']:>5}{c['lane']:>5}  {source_line(f, ln, roots)[:110]}", file=w)


if __name__ == "__main__":
    main()
