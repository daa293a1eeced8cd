"""A/B of the issue priority by work left (sdirt_psf.hip): one level per three passes left instead of per two."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioStep = 2;", "constexpr int kPrioStep = 3;")
