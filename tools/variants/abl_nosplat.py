"""Ablation (timing only, wrong PSFs): the splat of a ray reduced to one never-taken atomic."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_psf.hip",
    "        // the splat constants: one 64-byte scalar load per ray, dead again after the splat\n",
    "        if (ra > 2.0f) atomicAdd(&tl_[0], sx + sy + dx + dz);\n        return;\n")
