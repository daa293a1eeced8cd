"""Order experiment: as order_tight, and the open-ray ballot after the Newton step (last in the trip)."""
import sys
from _edit import sub
exec(open(__file__.replace("order_tight2", "order_tight")).read().split('import sys')[1].replace("from _edit import sub", ""))
sub(sys.argv[1], "sdirt_device.hpp",
    "        const unsigned long long open = __ballot(__builtin_fabsf(ft) > tol_loose);\n        const float tn = t - M::newton_step(ft, dfdt + eps);\n",
    "        const float tn = t - M::newton_step(ft, dfdt + eps);\n        const unsigned long long open = __ballot(__builtin_fabsf(ft) > tol_loose);\n")
