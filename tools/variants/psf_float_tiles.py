"""A/B partner of the double LDS tiles of k_psf_lr (round 5): float tiles (ds_add_f32) at every grid size, as rounds 1-4 had them."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "constexpr size_t kWideTilesMax = 39 * 1024;", "constexpr size_t kWideTilesMax = 0;")
