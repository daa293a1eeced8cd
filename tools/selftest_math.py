#!/usr/bin/env python3
"""Exhaustive verification of the lean division / square root (sdirt_device.hpp: Lean)
against the compiler's correctly rounded sequences, on the GPU.
  division: all 2^46 (mantissa_a, mantissa_b) pairs, operands in [1,2)   (~2.5 min)
  sqrt    : every fp32 with exponent in [-100, 128)                      (~0.5 s)
  sqrt_pos: every fp32 in [2^-100, 2^100]                                (~0.5 s)
Output is kept in profiles/rNN/selftest_math.txt."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd import _lib
from sdirt_amd.basics import dptr, stream_ptr

dev = torch.device("cuda:0")
h = _lib.lib()
out = torch.zeros(9, dtype=torch.int64, device=dev)


def run(mode, first, count, span=0):
    _lib.check(h.sdirt_selftest_math(mode, first, count, span, dptr(out), stream_ptr(dev)))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return int(o[0]), [hex(int(v) & 0xFFFFFFFFFFFFFFFF) for v in o[1:4]]


def pat(e):
    return (e + 127) << 23


t = time.time()
print("sqrt, every fp32 in [2^-100, inf):", run(0, pat(-100), 0x7F800000 - pat(-100)),
      f"{time.time() - t:.1f} s")
print("sqrt, +0 / +inf / NaN / -0 / negatives:", run(0, 0, 1), run(0, 0x7F800000, 1),
      run(0, 0x7FC00000, 1), run(0, 0x80000000, 1), run(0, 0x80800000, 0xFF800001 - 0x80800000))
t = time.time()
print("sqrt_pos (rsq + Markstein), every fp32 in [2^-100, 2^100]:", run(3, 0, 1 << 32),
      f"{time.time() - t:.1f} s")
t = time.time()
total = 0
step = 1 << 42
for k in range(0, 1 << 46, step):
    n, ex = run(2, k, step)
    total += n
    print(f"div, mantissa pairs [{k:#x}, {k + step:#x}): mismatches {n} {ex if n else ''}", flush=True)
print(f"div, ALL 2^46 mantissa pairs: mismatches {total}   ({time.time() - t:.0f} s)")
print("div, 2^38 random pairs, exponents within +-40, random signs:", run(1, 0, 1 << 38, 40))
