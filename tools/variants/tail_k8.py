"""A/B of the tail cut (sdirt_psf.hip: TailArgs): eight slices per tail point, a slice may be ONE pass of a workgroup."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr int kTailSlices = 4;", "constexpr int kTailSlices = 8;")
sub(sys.argv[1], "sdirt_psf.hip", "std::min<int64_t>(kTailSlices, S / (2 * kFused));", "std::min<int64_t>(kTailSlices, S / kFused);")
