"""A/B of the issue priority by work left (sdirt_psf.hip): level boundaries one pass lower (>= 6 / 4-5 / 2-3 / 1 passes left)."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioOffset = 0;", "constexpr int kPrioOffset = 1;")
