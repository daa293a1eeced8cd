#!/usr/bin/env python3
"""Static check of the hand-scheduled scalar prefetch in the trace loop (sdirt_device.hpp:
surf_issue / surf_wait) on the compiler's ISA.

surf_issue() starts three s_load_* into SGPRs and returns; surf_wait() is the s_waitcnt that
makes them readable.  The compiler does not know that the registers are "in flight" in between
(inline asm is opaque to its waitcnt insertion), so nothing may read or write them there -- in
particular no v_writelane spill of those SGPRs.  This script walks the control-flow graph of every
kernel in the `.s` from each issue site along ALL paths until an `s_waitcnt` that covers
lgkmcnt(0), and fails if an instruction on the way names one of the destination registers.

  python tools/check_prefetch_hazard.py sdirt_amd/csrc/obj/sdirt_psf-hip-amdgcn-amd-amdhsa-gfx950.s
(run by the compile rule of sdirt_amd/csrc/Makefile on the ISA of that very compile: an object of a
trace TU only exists if its ISA passed.)
"""
import re
import sys


def regs_of(tok):
    """'s[36:43]' -> {36..43}; 's12' -> {12}; else empty."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def sgprs_in(text):
    out = set()
    for tok in re.findall(r"(?<![\w.])s\[\d+:\d+\]|(?<![\w.\[])s\d+\b", text):
        out |= regs_of(tok)
    return out


def waits_lgkm0(ins):
    if not ins.startswith("s_waitcnt"):
        return False
    if "lgkmcnt(0)" in ins:
        return True
    m = re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", ins)      # raw immediate form
    return bool(m) and (int(m.group(1), 0) >> 8) & 0xF == 0


def check_kernel(name, lines):
    ins, label_at = [], {}
    for l in lines:
        t = l.split(";")[0].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    sites, errors = 0, []
    for i in range(len(ins) - 2):
        a, b, c = ins[i], ins[i + 1], ins[i + 2]
        if not (a.startswith("s_load_dwordx8") and b.startswith("s_load_dwordx4") and c.startswith("s_load_dword ")):
            continue
        if not re.search(r", 0x(20|30)$", b):
            continue
        if waits_lgkm0(ins[i + 3]):
            continue                      # load-and-wait form: nothing in flight afterwards
        sites += 1
        dst = set()
        for x in (a, b, c):
            dst |= regs_of(x.split()[1].rstrip(","))
        seen, work = set(), [i + 3]
        while work:
            j = work.pop()
            while j < len(ins) and j not in seen:
                seen.add(j)
                x = ins[j]
                if waits_lgkm0(x):
                    break
                used = sgprs_in(x.split(None, 1)[1] if " " in x else "")
                if used & dst:
                    errors.append(f"{name}: instruction {j} `{x}` touches in-flight s{sorted(used & dst)} "
                                  f"(issued at {i})")
                    break
                m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", x)
                if m:
                    work.append(label_at[m.group(1)])
                    # s_cbranch_execnz is LLVM's "always taken" form in uniform control flow (a wave
                    # that is running has exec != 0; the trace loop holds no divergent branch
                    # between issue and wait -- per-lane choices there are v_cndmask selects)
                    if x.startswith("s_branch") or x.startswith("s_cbranch_execnz"):
                        break
                if x.startswith("s_endpgm"):
                    break
                j += 1
    return sites, errors


def main(path):
    lines = open(path).read().splitlines()
    total, errors, i = 0, [], 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].strip().startswith("s_endpgm"):
                j += 1
            s, e = check_kernel(m.group(1)[:60], lines[i + 1:j + 1])
            total += s
            errors += e
            i = j
        i += 1
    print(f"check_prefetch_hazard: {total} prefetch sites, {len(errors)} hazards")
    for e in errors[:20]:
        print("  " + e)
    return 1 if errors or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
