"""A/B of the issue priority by work left (sdirt_psf.hip): one level per 1 pass(es) left instead of per quarter."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioStep = 2;", "constexpr int kPrioStep = 1;")
