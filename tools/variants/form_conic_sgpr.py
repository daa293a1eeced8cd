"""Form experiment (exact): the Newton loop's conic constants as SGPR operands instead of VGPR copies."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_device.hpp", "    using CV = ConicV;\n", "    using CV = ConicS;\n")
sub(root, "sdirt_device.hpp", "    const ConicV k = conic_v(s);\n", "    const ConicS k = conic_s(s);\n")
