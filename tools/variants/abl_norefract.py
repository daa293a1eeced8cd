"""Ablation (timing only): no refraction -- rays keep their direction (trip counts are table-driven,
so every Newton loop still runs the same number of trips)."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_device.hpp", "    refract<FWD, M>(s, pol, r);\n    return mask;", "    return mask;")
sub(sys.argv[1], "sdirt_device.hpp", "        if (s.do_refract()) refract<FWD, M>(s, NoPoly{}, r);\n", "")
