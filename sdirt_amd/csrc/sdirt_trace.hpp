// sdirt_trace.hpp -- the per-ray surface loop and its helpers, shared by the staged kernels
// (sdirt_trace.hip) and the fused PSF kernels (sdirt_psf.hip).
#pragma once
#include "sdirt_device.hpp"
#include "sdirt_host.hpp"

namespace sdirt {

__device__ __forceinline__ Ray load_ray(const sdirt_rays& R, int64_t i)
{
    Ray r;
    r.ox = R.ox[i]; r.oy = R.oy[i]; r.oz = R.oz[i];
    r.dx = R.dx[i]; r.dy = R.dy[i]; r.dz = R.dz[i];
    r.ra = R.ra[i];
    r.ob = R.obliq ? R.obliq[i] : 1.0f;
    return r;
}

__device__ __forceinline__ void store_ray(const sdirt_rays& R, int64_t i, const Ray& r)
{
    R.ox[i] = r.ox; R.oy[i] = r.oy; R.oz[i] = r.oz;
    R.dx[i] = r.dx; R.dy[i] = r.dy; R.dz[i] = r.dz;
    R.ra[i] = r.ra;
    if (R.obliq) R.obliq[i] = r.ob;
}

// *p |= m (p in LDS, m wave-uniform) by the first active lane of the wave.  Written out: the
// compiler's rendering of `if (lane == first) atomicOr(p, m)` goes through its wave-level atomic
// optimiser (mbcnt, two exec save/restore pairs, a readlane: ~20 instructions per surface).
__device__ __forceinline__ void lds_or_first_lane(uint32_t* p, uint32_t m)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p;
    uint64_t save, bit;
    uint32_t idx;
    asm volatile("s_ff1_i32_b64 %2, exec\n\t"
                 "s_lshl_b64 %1, 1, %2\n\t"
                 "s_and_saveexec_b64 %0, %1\n\t"
                 "ds_or_b32 %3, %4\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save), "=&s"(bit), "=&s"(idx) : "v"(addr), "v"(m) : "memory", "scc");
}

// Trace one ray through surfaces [first,last) in the travel direction.  The
// per-wave convergence masks are OR-ed into lds_mask[k] by the first active lane.
// PREFETCH = false: every surface's constants are loaded and waited for on the spot (the
// reference form the load-time selftest compares the prefetching loop against).
template <bool FWD, class M = Ieee, bool PREFETCH = true>
__device__ __forceinline__ void trace_ray(const DevSurface* __restrict__ lens, int first, int last,
                                          const void* trip_words, Ray& r, uint32_t* lds_mask)
{
    const int n = last - first;
    if (n <= 0) return;
    int k = FWD ? first : last - 1;
    SurfRaw cur;
    surf_issue<FWD>(cur, lens + k, trip_words, k);
    surf_wait(cur);
    for (int step = 0; step < n; ++step) {
        // the constants of the next surface (of this one again after the last: a harmless load)
        const int kn = step + 1 < n ? (FWD ? k + 1 : k - 1) : k;
        SurfRaw nxt;
        Surf s;
        s.a = cur.a; s.b = cur.b;
        const uint32_t m = surface_reaction<FWD, M>(s, lens + k, surf_trips(cur, k), r, [&] {
            if (PREFETCH) surf_issue<FWD>(nxt, lens + kn, trip_words, kn);
        });
        if (lds_mask && m != 0u) lds_or_first_lane(&lds_mask[k], m);
        if (!PREFETCH) surf_issue<FWD>(nxt, lens + kn, trip_words, kn);
        surf_wait(nxt);
        cur = nxt;
        k = kn;
    }
}

// byte `off` of this kernel's argument segment
__device__ __forceinline__ const void* kernarg_at(int off)
{
    return (const void*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}

template <class M = Ieee>
__device__ __forceinline__ Ray make_ray(float px, float py, float pz, float x2, float y2, float z2)
{
    Ray r;
    r.ox = px; r.oy = py; r.oz = pz;
    r.dx = x2 - px; r.dy = y2 - py; r.dz = z2 - pz;               // optics.py:490
    normalize3<M, true>(r.dx, r.dy, r.dz);                        // basics.py:245 (pupil != point)
    r.ra = 1.0f; r.ob = 1.0f;
    return r;
}

// One pupil sample: uniforms (theta / 2 pi, r^2 / R^2) -> a point of the pupil disc, optics.py:483-488.
// torch.cos / sin on the CPU (MKL VML) are < 1 ulp; the correctly rounded values, obtained here through fp64, agree
// with them far more often than a 1-2 ulp fp32 libm would, and that matters: d = o2 - o cancels against
// |o| ~ 1e4 mm.  O(spp) work, shared by all points -- cost is nil.
__device__ __forceinline__ void pupil_point(float u_theta, float u_r2, float pr2, float& x, float& y)
{
    const float theta = (u_theta * 2.0f) * (float)3.141592653589793;   // optics.py:483
    const float r = __builtin_sqrtf(u_r2 * pr2);                       // optics.py:484
    x = r * (float)__ocml_cos_f64((double)theta);
    y = r * (float)__ocml_sin_f64((double)theta);
}

__device__ __forceinline__ float block_max(float v, float* red)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float m = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
    return m;
}

}  // namespace sdirt
