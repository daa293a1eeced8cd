"""Form experiment (round 5): k_trace with an occupancy target of 8 waves per SIMD.  The staged trace kernel needs 40 VGPRs but 106
SGPRs (allocated as 112: 800 / 112 = 7 waves per SIMD); __launch_bounds__(256, 8) makes the compiler stay within 96."""
import sys
root = sys.argv[1]
p = root + "/sdirt_trace.hip"
s = open(p).read()
i = s.index("k_trace(TripTable trips /* kernarg offset 0 */")
j = s.rfind("__launch_bounds__(kBlock)", 0, i)
assert 0 < i - j < 120
s = s[:j] + "__launch_bounds__(kBlock, 8)" + s[j + len("__launch_bounds__(kBlock)"):]
open(p, "w").write(s)
