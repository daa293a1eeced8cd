"""Ablation (timing only): Newton's extra differentiable step and validity test (surfaces.py:563-586)
replaced by `t stays, valid = alive and t > 0`."""
import sys
from _edit import sub
root = sys.argv[1]
p = root + "/sdirt_device.hpp"
s = open(p).read()
a = s.index("    const float t1 = t - t0;   // :563")
b = s.index("    t_out = t;\n    return v;\n}", a)
s = s[:a] + "    const bool v = alive && t > 0.0f;\n" + s[b:]
open(p, "w").write(s)
