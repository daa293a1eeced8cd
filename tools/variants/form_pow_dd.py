"""Form experiment: the asphere's r2 ** n for n >= 4 from an fp32 double-float product chain (hi + lo, each step
hi' = RN(hi x), lo' = fma(hi, x, -hi') + lo x; the power = RN(hi + lo)) instead of the fp64 running product.  ~2^-44
relative before the final rounding: NOT always the bits of the fp64 form (a tie within 2^-44 rounds the other
way: ~3e-6 of the evaluations)."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_device.hpp", "    const double xd = (double)r2;\n    double accd = xd;                     // r2 ** n in fp64: ((x*x)*x)*...\n",
    "    float ph = r2, pl = 0.0f;             // r2 ** n as hi + lo\n")
sub(root, "sdirt_device.hpp", "            if (deg > 3) accd = accd * xd;\n            pw = n == 2 ? r2 * r2 : n == 3 ? (r2 * r2) * r2 : (float)accd;",
    "            if (deg > 3) { const float nh = ph * r2; pl = __builtin_fmaf(pl, r2, __builtin_fmaf(ph, r2, -nh)); ph = nh; }\n"
    "            pw = n == 2 ? r2 * r2 : n == 3 ? (r2 * r2) * r2 : ph + pl;")
