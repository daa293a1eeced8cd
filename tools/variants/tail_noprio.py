"""A/B of the end of a launch (sdirt_psf.hip): no issue priority by work left in the last generation."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kPrioLastGenerations = 1;", "constexpr int kPrioLastGenerations = 0;")
