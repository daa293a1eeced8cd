"""Ablation (timing only, NOT the same bits): the asphere's r2 ** n for n >= 4 as a plain fp32 running product
instead of the fp64 one -- the upper bound of what any cheaper exact form of those powers could save."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_device.hpp", "            if (deg > 3) accd = accd * xd;\n            pw = n == 2 ? r2 * r2 : n == 3 ? (r2 * r2) * r2 : (float)accd;",
    "            pw = n == 2 ? r2 * r2 : n == 3 ? (r2 * r2) * r2 : pw * r2;")
