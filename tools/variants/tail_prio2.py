"""A/B of the end of a launch (sdirt_psf.hip): the last TWO generations issue by work left."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioLastGenerations = 1;", "constexpr int kPrioLastGenerations = 2;")
