"""Form experiment (round 5): do workgroups that run in PHASE cost time?  One generation of 1024 identical workgroups takes
0.50 ms, each further generation 0.37 (tools/kbench.py --order same --n 1024 / 2048 / ...).  This variant delays the start of
the ray loop of a workgroup by k x SDIRT_STAG x 3.4 us, k = a 2-bit hash of blockIdx (the four workgroups that share a CU in
the first generation get different k if the dispatcher walks the CUs round-robin): build with -DSDIRT_STAG=<n>."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_psf.hip", '''    const float cx = (CENTER || from_parts) ? c_sh[0] : center[2 * n], cy = (CENTER || from_parts) ? c_sh[1] : center[2 * n + 1];''',
    '''#ifdef SDIRT_STAG
    {
        const int kst = (int)((blockIdx.x ^ (blockIdx.x >> 8) ^ (blockIdx.x >> 4)) & 3u);
        for (int i = 0; i < kst * SDIRT_STAG; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const float cx = (CENTER || from_parts) ? c_sh[0] : center[2 * n], cy = (CENTER || from_parts) ? c_sh[1] : center[2 * n + 1];''')
