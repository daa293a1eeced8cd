"""A/B of how a launch ends (sdirt_psf.hip: prio_by_work_left): no issue priority by work left -- round 5's kernels."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "return (int)std::max<int64_t>(0, blocks - 4ll * device_cus_or_default());", "return INT32_MAX;")
