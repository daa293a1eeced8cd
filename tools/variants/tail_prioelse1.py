"""A/B of the end of a launch (sdirt_psf.hip): workgroups before the last generation at constant issue priority 1."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioElse = 0;", "constexpr int kPrioElse = 1;")
