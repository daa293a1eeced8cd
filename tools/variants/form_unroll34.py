"""Form experiment (exact): tables of exactly 3 / 4 trips (the common ones) run straight-line code."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_device.hpp",
    "        while (left > 0) trip(std::false_type{});\n",
    "        if (!adaptive && cap == 3) { trip(std::false_type{}); trip(std::false_type{}); trip(std::false_type{}); }\n"
    "        else if (!adaptive && cap == 4) { trip(std::false_type{}); trip(std::false_type{}); trip(std::false_type{}); trip(std::false_type{}); }\n"
    "        while (left > 0) trip(std::false_type{});\n")
