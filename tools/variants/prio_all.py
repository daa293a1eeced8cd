"""A/B of how a launch ends (sdirt_psf.hip): EVERY workgroup issues by work left, not only the last generation."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip", "return (int)std::max<int64_t>(0, blocks - 4ll * device_cus_or_default());", "return 0;")
