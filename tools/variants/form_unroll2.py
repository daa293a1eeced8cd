"""Form experiment (exact): the short Newton loop (tables of <= 4 trips) unrolled by two when the
trip count is the host's (not adaptive) -- half the loop branches and counter tests."""
import sys
from _edit import sub
sub(sys.argv[1], "sdirt_device.hpp",
    "        while (left > 0) trip(std::false_type{});\n",
    "        if (!adaptive)\n"
    "            while (left > 1) { trip(std::false_type{}); trip(std::false_type{}); }\n"
    "        while (left > 0) trip(std::false_type{});\n")
