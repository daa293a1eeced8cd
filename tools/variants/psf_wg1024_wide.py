"""Form experiment (second A/B): k_psf_lr with 1024-thread workgroups, two per CU (the same 8 waves
per SIMD), so that DOUBLE tiles fit at ks 65 as well (2 x 67.6 KB)."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_host.hpp", "constexpr int kFused = 512;", "constexpr int kFused = 1024;")
sub(root, "sdirt_psf.hip", "constexpr size_t kWideTilesMax = 39 * 1024;", "constexpr size_t kWideTilesMax = 79 * 1024;")
