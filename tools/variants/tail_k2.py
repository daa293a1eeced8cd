"""A/B of the tail cut (sdirt_psf.hip: TailArgs): two slices per tail point instead of four."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kTailSlices = 4;", "constexpr int kTailSlices = 2;")
