"""A/B of the tail cut (sdirt_psf.hip: TailArgs): half a generation of tail points (2 per CU) instead of a whole one."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "tp.n_tail = (int)slots; tp.K = K;", "tp.n_tail = (int)(slots / 2); tp.K = K;")
