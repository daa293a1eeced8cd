"""Form experiment (round 5): the wave-per-pixel renderers with an occupancy target in __launch_bounds__ --
k_local_psf_render_wave at 8 waves per SIMD (<= 64 VGPRs; 68 today = 7 waves: 1792 workgroup slots for the 6144 workgroups of a
512 x 768 frame = 3.43 generations; 8 waves = 2048 slots = 3.0), k_psfnet_render_wave at 6 (<= 80 VGPRs; 88 today = 5 waves)."""
import sys
root = sys.argv[1]
p = root + "/sdirt_render.hip"
s = open(p).read()
for name, waves in (("k_local_psf_render_wave(const float* __restrict__ img", 8), ("k_psfnet_render_wave(const float* __restrict__ img", int(sys.argv[2]) if len(sys.argv) > 2 else 6)):
    i = s.index(name)
    j = s.rfind("__launch_bounds__(kBlock)", 0, i)
    assert i - j < 200, name
    s = s[:j] + f"__launch_bounds__(kBlock, {waves})" + s[j + len("__launch_bounds__(kBlock)"):]
open(p, "w").write(s)
