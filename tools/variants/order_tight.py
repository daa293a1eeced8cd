"""Order experiment (same arithmetic): z of the trial point computed where it is used (next to ft) instead of
with x and y at the top of the trip."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_device.hpp",
    "        const float nx = r.ox + r.dx * t, ny = r.oy + r.dy * t, nz = r.oz + r.dz * t;\n        const float rr = nx * nx + ny * ny;\n        // g(x*valid, y*valid)",
    "        const float nx = r.ox + r.dx * t, ny = r.oy + r.dy * t;\n        const float rr = nx * nx + ny * ny;\n        // g(x*valid, y*valid)")
sub(sys.argv[1], "sdirt_device.hpp",
    "        const float ft = (g + k.d) - nz;\n        const float dfdt = M::dfdt(dgd, dd * t + dox, r.dz);            // dgd * dr2dt",
    "        const float nz = r.oz + r.dz * t;\n        const float ft = (g + k.d) - nz;\n        const float dfdt = M::dfdt(dgd, dd * t + dox, r.dz);            // dgd * dr2dt")
