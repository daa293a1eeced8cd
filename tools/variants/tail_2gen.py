"""A/B of the tail cut (sdirt_psf.hip: TailArgs): two generations of tail points where the launch has them."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "tp.n_tail = (int)slots; tp.K = K;", "tp.n_tail = (int)std::min<int64_t>(N, 2 * slots); tp.K = K;")
