"""Form experiment (exact, same results): the three direction selects of refract() through vcc
(VOP2 v_cndmask_b32_e32, full rate) instead of an SGPR-pair mask (VOP3, half rate), and eta as a
VGPR copy (three uses)."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_device.hpp",
    "    const float eta = s.eta(), eta2 = s.eta2();\n",
    "    const float eta = to_vgpr(s.eta()), eta2 = s.eta2();\n")
sub(root, "sdirt_device.hpp",
    "    ndx = v ? ndx : r.dx; ndy = v ? ndy : r.dy; ndz = v ? ndz : r.dz;\n",
    "    {\n"
    "        const unsigned long long vm = __ballot(v);\n"
    "        asm volatile(\"s_mov_b64 vcc, %3\\n\\t\"\n"
    "                     \"v_cndmask_b32_e32 %0, %4, %0, vcc\\n\\t\"\n"
    "                     \"v_cndmask_b32_e32 %1, %5, %1, vcc\\n\\t\"\n"
    "                     \"v_cndmask_b32_e32 %2, %6, %2, vcc\"\n"
    "                     : \"+v\"(ndx), \"+v\"(ndy), \"+v\"(ndz) : \"s\"(vm), \"v\"(r.dx), \"v\"(r.dy), \"v\"(r.dz) : \"vcc\");\n"
    "    }\n")
