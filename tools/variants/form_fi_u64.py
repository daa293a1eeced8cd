"""Form experiment (timing and accuracy of the idea, not a product path): k_forward_integral_tiles' float64 accumulators
as 64-bit FIXED-POINT ones -- contribution * 2^32 rounded to an integer, ds_add_u64 (8-21 LDS cycles per wave instruction
against 17-33 for ds_add_f64), converted back on the way out.  Valid as it stands for weights in [0, 1] only (no scale per point)."""
import sys
from _edit import sub
root = sys.argv[1]
p = root + "/sdirt_psf.hip"
s = open(p).read()
a = s.index("template <bool HAVE_R, bool BIG, class ACC, class M>")
b = s.index("static void launch_normalize")
k = s[a:b]
for name in ("tl_", "trr"):
    for tap in ("tl", "tr", "bl", "br"):
        sl = "sl" if name == "tl_" else "sr"
        old = f"atomicAdd(&{name}[tp.i_{tap}], (ACC)(tp.w_{tap} * {sl}));"
        new = (f"fi_add(&{name}[tp.i_{tap}], tp.w_{tap} * {sl});")
        assert k.count(old) == 1, old
        k = k.replace(old, new)
k = k.replace("Lg[e] = (float)src[e];", "Lg[e] = fi_out(src[e]);").replace("if (HAVE_R) Rg[e] = (float)src[tile + e];", "if (HAVE_R) Rg[e] = fi_out(src[tile + e]);")
k = k.replace("const float a = (float)src[e];", "const float a = fi_out(src[e]);").replace("const float c = (float)src[tile + e];", "const float c = fi_out(src[tile + e]);")
helpers = '''
__device__ __forceinline__ void fi_add(double* p, float c)
{
    const long long v = __double2ll_rn((double)c * 4294967296.0);
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
__device__ __forceinline__ void fi_add(float* p, float c) { atomicAdd(p, c); }
__device__ __forceinline__ float fi_out(double v) { return (float)((double)__double_as_longlong(v) * (1.0 / 4294967296.0)); }
__device__ __forceinline__ float fi_out(float v) { return v; }
'''
s = s[:a] + helpers + k + s[b:]
open(p, "w").write(s)
