// sdirt_dp.hip -- kernels and C ABI of libsdirt_dp.so (MI355X / gfx950 only).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// (sdirt_amd/csrc/Makefile).  See include/sdirt_dp.h for the ABI and DESIGN.md
// for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/sdirt_dp.h"
#include "sdirt_device.hpp"

#include "sdirt_host.hpp"

using namespace sdirt;

// ---------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------
struct sdirt_lens {
    int32_t n_surfaces;
    DevSurface* dev;               // device table [n_surfaces]
    std::vector<DevSurface> host;  // host mirror
};

// Per-surface constant block; every double->float rounding happens here, at the
// same place the reference's torch scalar handling performs it.
static DevSurface make_dev_surface(const sdirt_surface_desc& in)
{
    DevSurface s;
    std::memset(&s, 0, sizeof(s));
    SurfHot& h = s.h;
    const int deg = in.kind == SDIRT_ASPHERE ? in.ai_degree : 0;
    h.d = in.d; h.c = in.c; h.k = in.k;
    h.r2_lim = (float)(in.r * in.r);
    h.c2 = in.c * in.c;
    h.onepk = 1.0f + in.k;
    if (in.kind != SDIRT_PLANE) {
        float rc = 1.0f / h.c2;                       // tensor.reciprocal()
        rc = rc * (float)(1.0 - 1e-9);                // * python float (1-EPSILON)
        h.lim_loose = rc / h.onepk;
        h.d_plus_R = in.d + 1.0f / in.c;
        h.d_plus_R_b = h.d_plus_R;
        h.lim_tight = in.k > -1.0f ? std::min(h.r2_lim, h.lim_loose) : h.r2_lim;
    } else {
        h.lim_tight = (float)in.r;                    // planes: the aperture radius itself
    }
    const double eta_f = in.n1 / in.n2, eta_b = in.n2 / in.n1;
    h.eta_f = (float)eta_f;  h.eta2_f = (float)(eta_f * eta_f);
    h.eta_b = (float)eta_b;  h.eta2_b = (float)(eta_b * eta_b);
    const bool do_refract = in.kind == SDIRT_PLANE ? (eta_f != 1.0) : true;
    h.flags = (uint32_t)in.kind | (do_refract ? kFlagRefract : 0u) | (in.k > -1.0f ? kFlagKgtM1 : 0u) |
              (in.c > 0.0f ? kFlagCpos : 0u) | (h.onepk == 1.0f ? kFlagUnitK : 0u) | ((uint32_t)deg << 8);
    for (int i = 0; i < kMaxAi; ++i) {
        s.p.ai[i] = i < deg ? in.ai[i] : 0.0f;
        s.p.kai[i] = (float)(i + 1) * s.p.ai[i];
    }
    return s;
}

static DevDpParams make_dp(const sdirt_dp_params* dp)
{
    DevDpParams p;
    const double h = dp ? dp->h : 0.78, f = dp ? dp->f : 1.44, w = dp ? dp->w : 0.3,
                 r = dp ? dp->r : 0.5;
    p.h = (float)h; p.f = (float)f; p.w = (float)w; p.r = (float)r;
    p.fmh = (float)(f - h);
    p.rr = p.r * p.r;
    p.big = r > 0.5;
    p.have_r = dp != nullptr;
    int ex = 0;
    p.r_pow2 = std::frexp(p.r, &ex) == 0.5f;
    p.inv_r = 1.0f / p.r;
    p.tr = std::asin((1.0f / p.r) * 0.5f);
    p.tl = (float)3.141592653589793 - p.tr;
    return p;
}

static SplatGeom make_geom(double ps, int ks)
{
    SplatGeom g;
    const double hi = (ks / 2.0 - 0.5) * ps, lo = (-ks / 2.0 + 0.5) * ps;
    g.lim = (float)(hi - 0.01 * ps);
    g.x_min = (float)lo;
    g.y_max = (float)hi;
    g.dx_rng = (float)(hi - lo);
    g.dy_rng = (float)(lo - hi);
    g.ksm1 = (float)(ks - 1);
    g.ks = ks;
    return g;
}

// Everything the splat of one ray needs (window geometry + dual-pixel parameters), as ONE
// 64-byte block at offset 0 of k_psf_lr's kernel-argument segment: the kernel fetches it with one
// scalar load per RAY, right before the splat, instead of keeping ~20 SGPRs alive through the
// trace (where round 1's build parked them in VGPR lanes: v_writelane / v_readlane traffic).
struct alignas(64) SplatBlock {
    float lim, x_min, y_max, dx_rng, dy_rng, ksm1;
    int32_t ks;
    float h, f, w, r, fmh, rr, inv_r;
    int32_t r_pow2;
    int32_t pad;
};
static_assert(sizeof(SplatBlock) == 64, "layout");

static SplatBlock make_splat_block(const SplatGeom& g, const DevDpParams& p)
{
    SplatBlock b;
    b.lim = g.lim; b.x_min = g.x_min; b.y_max = g.y_max; b.dx_rng = g.dx_rng; b.dy_rng = g.dy_rng;
    b.ksm1 = g.ksm1; b.ks = g.ks;
    b.h = p.h; b.f = p.f; b.w = p.w; b.r = p.r; b.fmh = p.fmh; b.rr = p.rr; b.inv_r = p.inv_r;
    b.r_pow2 = p.r_pow2; b.pad = 0;
    return b;
}

// Newton trip counts of one launch, one signed byte per surface, passed by value at a FIXED
// offset of the kernel-argument segment; the kernels read the dword of surface k from there
// with a scalar load that shares the round trip of the surface's constant block (a byte-indexed
// by-value table made the compiler issue a vector load and wait for it once per surface).
struct alignas(64) TripTable {
    uint32_t w[SDIRT_MAX_SURFACES / 4];
};
static_assert(sizeof(TripTable) == 64, "layout");

static int make_trips(const sdirt_lens* lens, const int32_t* trips, TripTable& tt)
{
    for (int k = 0; k < SDIRT_MAX_SURFACES / 4; ++k) tt.w[k] = 0;
    for (int k = 0; k < lens->n_surfaces; ++k) {
        int v = trips ? trips[k] : SDIRT_NEWTON_MAXITER;
        if (v < -SDIRT_NEWTON_MAXITER || v > SDIRT_NEWTON_MAXITER)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "trips[%d]=%d outside [-%d,%d]", k, v,
                        SDIRT_NEWTON_MAXITER, SDIRT_NEWTON_MAXITER);
        tt.w[k >> 2] |= (uint32_t)(uint8_t)(int8_t)v << ((k & 3) * 8);
    }
    return SDIRT_OK;
}

// ---------------------------------------------------------------------------
// device helpers shared by the kernels
// ---------------------------------------------------------------------------
constexpr int kBlock = 256;
// Workgroup size of the two fused kernels.  512 threads = 8 waves share one pair of
// L/R tiles (33.8 KB at ks 65): 4 workgroups = 32 waves per CU = 8 per SIMD; both
// kernels need <= 43 VGPRs.  Measured on config 2 (tools/kbench.py): k_psf_lr 12.69 ms
// at 256 threads (LDS-limited to 4 waves/SIMD), 11.76 ms at 512, 15.7 ms at 1024.
#ifndef SDIRT_FUSED_BLOCK
#define SDIRT_FUSED_BLOCK 512
#endif
constexpr int kFused = SDIRT_FUSED_BLOCK;

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
template <bool FWD, class M = Ieee>
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
        const uint32_t m = surface_reaction<FWD, M>(s, lens + k, surf_trips(cur, k), r,
                                                    [&] { surf_issue<FWD>(nxt, lens + kn, trip_words, kn); });
        if (lds_mask && m != 0u) lds_or_first_lane(&lds_mask[k], m);
        surf_wait(nxt);
        cur = nxt;
        k = kn;
    }
}

#ifdef SDIRT_SPEC_HEADER
#include SDIRT_SPEC_HEADER        // per-prescription specialisation (tools/spec_build.py): SpecA*/SpecC* and trace_spec
#endif

// byte `off` of this kernel's argument segment
__device__ __forceinline__ const void* kernarg_at(int off)
{
    return (const void*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}

// ---------------------------------------------------------------------------
// staged kernels
// ---------------------------------------------------------------------------
__global__ void k_points_to_object(const float* __restrict__ pts, int64_t N, float tf, float rl,
                                   float sw, float sh, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float depth = pts[3 * i + 2];
    const float scale = ((-depth) * tf) / rl;                 // optics.py:1305
    out[3 * i] = ((pts[3 * i] * scale) * sw) / 2.0f;          // optics.py:959
    out[3 * i + 1] = ((pts[3 * i + 1] * scale) * sh) / 2.0f;  // optics.py:960
    out[3 * i + 2] = depth;
}

__global__ void k_pupil_samples(const float* __restrict__ ut, const float* __restrict__ ur,
                                int64_t S, float pr2, float* __restrict__ x2,
                                float* __restrict__ y2)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float theta = (ut[s] * 2.0f) * (float)3.141592653589793;   // optics.py:483
    const float r = __builtin_sqrtf(ur[s] * pr2);                     // optics.py:484
    // torch.cos/sin on CPU (MKL VML) are <1 ulp; the correctly rounded values,
    // obtained here through fp64, agree with them far more often than a 1-2 ulp
    // fp32 libm would, and that matters: d = o2 - o cancels against |o| ~ 1e4 mm
    // (DESIGN.md §5).  O(spp) work, shared by all points -- cost is nil.
    x2[s] = r * (float)__ocml_cos_f64((double)theta);
    y2[s] = r * (float)__ocml_sin_f64((double)theta);
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

__global__ void k_sample_rays(const float* __restrict__ po, int64_t N, const float* __restrict__ x2,
                              const float* __restrict__ y2, int64_t S, float pz, sdirt_rays R)
{
    const int64_t M = S * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / N, n = i - s * N;
        Ray r = make_ray(po[3 * n], po[3 * n + 1], po[3 * n + 2], x2[s], y2[s], pz);
        store_ray(R, i, r);
    }
}

__global__ void k_rays_from_aos(const float* __restrict__ o, const float* __restrict__ d,
                                const float* __restrict__ ra, int64_t M, int normalize, sdirt_rays R)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        Ray r;
        r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
        r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
        if (normalize) normalize3<Ieee>(r.dx, r.dy, r.dz);
        r.ra = ra ? ra[i] : 1.0f;
        r.ob = 1.0f;
        store_ray(R, i, r);
    }
}

__global__ void k_rays_to_aos(sdirt_rays R, int64_t M, float* __restrict__ o, float* __restrict__ d)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (o) { o[3 * i] = R.ox[i]; o[3 * i + 1] = R.oy[i]; o[3 * i + 2] = R.oz[i]; }
        if (d) { d[3 * i] = R.dx[i]; d[3 * i + 1] = R.dy[i]; d[3 * i + 2] = R.dz[i]; }
    }
}

template <bool FWD, class MP>
__global__ void __launch_bounds__(kBlock)
k_trace(TripTable trips /* kernarg offset 0 */, const DevSurface* __restrict__ lens, int K, int first,
        int last, sdirt_rays R, int64_t M, uint32_t* __restrict__ conv_mask)
{
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        Ray r = load_ray(R, i);
        trace_ray<FWD, MP>(lens, first, last, kernarg_at(0), r, conv_mask ? lds_mask : nullptr);
        store_ray(R, i, r);
    }
    __syncthreads();
    if (conv_mask && (int)threadIdx.x < K && lds_mask[threadIdx.x])
        atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
}

__global__ void k_propagate(float z, sdirt_rays R, int64_t M)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float dz = R.dz[i];
        const float t = (z - R.oz[i]) / dz;
        R.ox[i] = R.ox[i] + R.dx[i] * t;
        R.oy[i] = R.oy[i] + R.dy[i] * t;
        R.oz[i] = R.oz[i] + dz * t;
    }
}

// Centroid over the spp axis with fp64 accumulation; one thread per point so
// that consecutive lanes read consecutive addresses of the [S,N] arrays.
__global__ void k_center_from_rays(sdirt_rays R, int64_t S, int64_t N, float* __restrict__ center,
                                   int32_t* __restrict__ any_valid)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int any = 0;
    for (int64_t s = 0; s < S; ++s) {
        const int64_t i = s * N + n;
        const float ra = R.ra[i];
        sx += (double)(R.ox[i] * ra);
        sy += (double)(R.oy[i] * ra);
        sr += (double)ra;
        any |= (ra == 1.0f);
    }
    const float den = (float)sr + (float)1e-9;
    center[2 * n] = -((float)sx / den);
    center[2 * n + 1] = -((float)sy / den);
    if (any_valid && any) atomicOr(any_valid, 1);
}

// forward_integral on SoA [S,N] rays: one thread per ray (coalesced reads),
// contributions added to the pre-zeroed [N,ks,ks] grids with global float
// atomics -- consecutive lanes are consecutive POINTS, so the 64 atomics of a
// wave instruction go to 64 different tiles.
__global__ void __launch_bounds__(kBlock)
k_forward_integral(sdirt_rays R, int64_t S, int64_t N, SplatGeom gm, DevDpParams dp,
                   const float* __restrict__ center, float* __restrict__ lg,
                   float* __restrict__ rg)
{
    const int64_t M = S * N;
    const int64_t tile = (int64_t)gm.ks * gm.ks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i % N;
        SplatTaps tp;
        if (!splat_taps(gm, R.ox[i], R.oy[i], center[2 * n], center[2 * n + 1], R.ra[i], tp))
            continue;
        const float x_tan = (-R.dx[i]) / R.dz[i];    // monte_carlo.py:48
        float sl, sr;
        if (dp.big) dp_weights_big(dp, x_tan, sl, sr);
        else dp_weights_small(dp, x_tan, sl, sr);
        float* L = lg + n * tile;
        atomicAdd(L + tp.i_tl, tp.w_tl * sl);
        atomicAdd(L + tp.i_tr, tp.w_tr * sl);
        atomicAdd(L + tp.i_bl, tp.w_bl * sl);
        atomicAdd(L + tp.i_br, tp.w_br * sl);
        if (rg && dp.have_r) {
            float* Rr = rg + n * tile;
            atomicAdd(Rr + tp.i_tl, tp.w_tl * sr);
            atomicAdd(Rr + tp.i_tr, tp.w_tr * sr);
            atomicAdd(Rr + tp.i_bl, tp.w_bl * sr);
            atomicAdd(Rr + tp.i_br, tp.w_br * sr);
        }
    }
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

// optics.py:983-987: one workgroup per point.
__global__ void __launch_bounds__(kBlock) k_psf_normalize(float* __restrict__ psf, int tile)
{
    __shared__ float red[kBlock / 64];
    float* g = psf + (int64_t)blockIdx.x * tile;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, g[i]);
    mx = block_max(mx, red);
    const float den = mx + 1e-6f;
    for (int i = threadIdx.x; i < tile; i += blockDim.x) g[i] = g[i] / den;
}

// ---------------------------------------------------------------------------
// fused kernels
// ---------------------------------------------------------------------------

// psf_center: one workgroup per point, Sc rays, fp64 partial sums reduced in a
// fixed order (deterministic).
template <class HotMath>
__global__ void __launch_bounds__(kFused, 8)
k_chief_center(TripTable trips /* kernarg offset 0 */, const DevSurface* __restrict__ lens, int K,
               const float* __restrict__ po, const float* __restrict__ xc,
               const float* __restrict__ yc, int Sc, float pz, float zs,
               float* __restrict__ center, int32_t* __restrict__ any_valid,
               uint32_t* __restrict__ conv_mask)
{
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    __shared__ double red[3][kFused];
    __shared__ int red_any;
    const int n = blockIdx.x;
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    if (threadIdx.x == 0) red_any = 0;
    __syncthreads();
    const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int any = 0;
    for (int s = threadIdx.x; s < Sc; s += blockDim.x) {
        Ray r = make_ray<HotMath>(px, py, pzo, xc[s], yc[s], pz);
        trace_ray<true, HotMath>(lens, 0, K, kernarg_at(0), r, conv_mask ? lds_mask : nullptr);
        propagate_to<HotMath>(r, zs);
        sx += (double)(r.ox * r.ra);
        sy += (double)(r.oy * r.ra);
        sr += (double)r.ra;
        any |= (r.ra == 1.0f);
    }
    red[0][threadIdx.x] = sx; red[1][threadIdx.x] = sy; red[2][threadIdx.x] = sr;
    if (any) red_any = 1;
    __syncthreads();
    for (int off = kFused / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
            red[2][threadIdx.x] += red[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float den = (float)red[2][0] + (float)1e-9;
        center[2 * n] = -((float)red[0][0] / den);
        center[2 * n + 1] = -((float)red[1][0] / den);
        if (any_valid && red_any) atomicOr(any_valid, 1);
    }
    if (conv_mask && (int)threadIdx.x < K && lds_mask[threadIdx.x])
        atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
}

// psf_diff fused: sample -> trace -> propagate -> window -> DP weights -> LDS
// splat -> (max-normalise) -> store.  gridDim.x = N * nsplit; the workgroup
// (n, j) handles samples [j*chunk, (j+1)*chunk) of point n.
//   nsplit == 1 : the tile is complete in LDS -> normalise (flag) and store.
//   nsplit  > 1 : tiles are added to the pre-zeroed output with global float
//                 atomics; the caller normalises afterwards.
// Arguments of the optional in-kernel chief-ray pass (CENTER instantiations, nsplit == 1): the
// workgroup first traces the Sc shrunk-pupil samples of its point through the GREEN lens table
// (optics.py:900), reduces the centroid exactly like k_chief_center, and only then splats.
struct CenterArgs {
    const DevSurface* lens_c;
    const float* xc;
    const float* yc;
    int Sc;
    float* center_out;       // [N,2]
    int32_t* any_valid;
    uint32_t* conv_mask_c;
};

// __launch_bounds__(512, 8): four workgroups per CU = 8 waves per SIMD (<= 80 SGPRs, <= 64 VGPRs).
// The big-radius microlens branch (corner-clipped areas, monte_carlo.py:242-372) needs ~90 VGPRs:
// capped at 64 it would spill 60 of them to scratch, so it runs at 4 waves per SIMD instead.
// `sb` MUST stay the first parameter: the kernel reads it as a 64-byte block at offset 0 of its
// kernel-argument segment (see SplatBlock) and never through the parameter itself.
// blockIdx.y = wavelength slot w of a multi-wavelength launch (psf_rgb: gridDim.y = 3, one lens
// table, pupil sample set, trip table and mask row per slot; a plain call has gridDim.y = 1): the
// output is [N, gridDim.y, ks, ks], the reference's psf_rgb layout (optics.py:1015).
struct LensSet {
    const DevSurface* p[SDIRT_MAX_WAVELENGTHS];
};
struct TripSet {
    TripTable t[SDIRT_MAX_WAVELENGTHS];
};
#ifdef SDIRT_DBG_CLOCK
__device__ unsigned long long g_dbg_clock[4];
#endif
template <bool HAVE_R, bool BIG, class HotMath, bool CENTER>
__global__ void __launch_bounds__(kFused, BIG ? 4 : 8)
k_psf_lr(SplatBlock sb /* kernarg offset 0 */, TripSet trips /* 64 */, TripSet trips_c /* 64 + 64 W */,
         LensSet lens_set, int K, const float* __restrict__ po, const float* __restrict__ x2,
         const float* __restrict__ y2, int S, int nsplit, int chunk, float pz, float zs, int ks, float tr,
         float tl, const float* __restrict__ center, uint32_t flags, float* __restrict__ lout,
         float* __restrict__ rout, uint32_t* __restrict__ conv_mask, CenterArgs ca)
{
    extern __shared__ __attribute__((aligned(16))) float tiles[];   // [L | R] ks*ks each
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    __shared__ float red[kFused / 64];
    __shared__ float c_sh[2];
    const int tile = ks * ks;
    float* tl_ = tiles;
    float* trr = tiles + tile;
    const int n = blockIdx.x / nsplit;
    const int j = blockIdx.x - n * nsplit;
    const int w = blockIdx.y, W = gridDim.y;
    const int N = gridDim.x / nsplit;
    const DevSurface* __restrict__ lens = lens_set.p[w];
    constexpr int kTripsAt = 64, kTripsCAt = 64 + 64 * SDIRT_MAX_WAVELENGTHS;
    x2 += (int64_t)w * S; y2 += (int64_t)w * S;
    if (conv_mask) conv_mask += w * SDIRT_MAX_SURFACES;
#ifdef SDIRT_SPEC_HEADER
    const u32x16 spec_tw = sload_block(kernarg_at(kTripsAt + 64 * w));       // both trip tables, once per workgroup
    const u32x16 spec_tw_c = sload_block(kernarg_at(kTripsCAt + 64 * w));
#endif
    if (CENTER) {
        ca.xc += (int64_t)w * ca.Sc; ca.yc += (int64_t)w * ca.Sc;
        ca.center_out += (int64_t)w * N * 2;
        if (ca.any_valid) ca.any_valid += w;
        if (ca.conv_mask_c) ca.conv_mask_c += w * SDIRT_MAX_SURFACES;
    } else if (center) {
        center += (int64_t)w * N * 2;
    }
    const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];
#ifdef SDIRT_DBG_CLOCK
    const unsigned long long dbg_t0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    if (CENTER) {
        // ---- chief-ray centre of this point (same arithmetic and reduction order as
        // k_chief_center; the fp64 scratch aliases the not-yet-used tile memory)
        double* redd = reinterpret_cast<double*>(tiles);             // [3][kFused]
        if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
        if (threadIdx.x == 0) c_sh[0] = 0.0f;
        __syncthreads();
        double sx = 0.0, sy = 0.0, sr = 0.0;
        int any = 0;
        for (int s = threadIdx.x; s < ca.Sc; s += blockDim.x) {
            Ray r = make_ray<HotMath>(px, py, pzo, ca.xc[s], ca.yc[s], pz);
#ifdef SDIRT_SPEC_HEADER
            trace_spec<true, HotMath>(spec_tw_c, r, ca.conv_mask_c ? lds_mask : nullptr);
#else
            trace_ray<true, HotMath>(ca.lens_c, 0, K, kernarg_at(kTripsCAt + 64 * w), r,
                                     ca.conv_mask_c ? lds_mask : nullptr);
#endif
            propagate_to<HotMath>(r, zs);
            sx += (double)(r.ox * r.ra);
            sy += (double)(r.oy * r.ra);
            sr += (double)r.ra;
            any |= (r.ra == 1.0f);
        }
        redd[threadIdx.x] = sx; redd[kFused + threadIdx.x] = sy; redd[2 * kFused + threadIdx.x] = sr;
        if (any) c_sh[0] = 1.0f;                                     // benign race: all write 1
        __syncthreads();
        for (int off = kFused / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                redd[threadIdx.x] += redd[threadIdx.x + off];
                redd[kFused + threadIdx.x] += redd[kFused + threadIdx.x + off];
                redd[2 * kFused + threadIdx.x] += redd[2 * kFused + threadIdx.x + off];
            }
            __syncthreads();
        }
        if (ca.conv_mask_c && (int)threadIdx.x < K && lds_mask[threadIdx.x])
            atomicOr(&ca.conv_mask_c[threadIdx.x], lds_mask[threadIdx.x]);
        const float any_f = c_sh[0];
        const float den = (float)redd[2 * kFused] + (float)1e-9;
        const float ccx = -((float)redd[0] / den), ccy = -((float)redd[kFused] / den);
        __syncthreads();                                             // everyone has read redd / c_sh
        if (threadIdx.x == 0) {
            ca.center_out[2 * n] = ccx; ca.center_out[2 * n + 1] = ccy;
            c_sh[0] = ccx; c_sh[1] = ccy;
            if (ca.any_valid && any_f != 0.0f) atomicOr(ca.any_valid, 1);
        }
    }

    for (int i = threadIdx.x; i < (HAVE_R ? 2 : 1) * tile; i += blockDim.x) tiles[i] = 0.0f;
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    __syncthreads();

    const float cx = CENTER ? c_sh[0] : center[2 * n], cy = CENTER ? c_sh[1] : center[2 * n + 1];
    const int s_end = min(S, (j + 1) * chunk);
    const void* kernarg = kernarg_at(0);
    auto splat = [&](float sx, float sy, float dx, float dz, float ra) {
#ifdef SDIRT_ABL_NOSPLAT
        if (ra > 2.0f) atomicAdd(&tl_[0], sx + sy + dx + dz);
        return;
#endif
        // the splat constants: one 64-byte scalar load per ray, dead again after the splat
        const u32x16 q = sload_block(kernarg);
        const auto F = [&](int i) { return __uint_as_float(q[i]); };
        SplatGeom gm;
        gm.lim = F(0); gm.x_min = F(1); gm.y_max = F(2); gm.dx_rng = F(3); gm.dy_rng = F(4);
        gm.ksm1 = F(5); gm.ks = (int)q[6];
        DevDpParams dp;
        dp.h = F(7); dp.f = F(8); dp.w = F(9); dp.r = F(10); dp.fmh = F(11); dp.rr = F(12);
        dp.inv_r = F(13); dp.r_pow2 = (int)q[14]; dp.tr = tr; dp.tl = tl; dp.big = BIG; dp.have_r = HAVE_R;
        SplatTaps tp;
        if (!splat_taps(gm, UDiv<HotMath>::make(gm.dy_rng), UDiv<HotMath>::make(gm.dx_rng), sx, sy, cx, cy,
                        ra, tp))
            return;
        const float x_tan = HotMath::div(-dx, dz);
        float sl, sr;
        if (BIG) dp_weights_big(dp, x_tan, sl, sr);      // separate instantiation: the rarely
        else dp_weights_small(dp, UDiv<HotMath>::make(dp.fmh), x_tan, sl, sr);   // used r > 0.5 branch costs registers
        atomicAdd(&tl_[tp.i_tl], tp.w_tl * sl);
        atomicAdd(&tl_[tp.i_tr], tp.w_tr * sl);
        atomicAdd(&tl_[tp.i_bl], tp.w_bl * sl);
        atomicAdd(&tl_[tp.i_br], tp.w_br * sl);
        if (HAVE_R) {
            atomicAdd(&trr[tp.i_tl], tp.w_tl * sr);
            atomicAdd(&trr[tp.i_tr], tp.w_tr * sr);
            atomicAdd(&trr[tp.i_bl], tp.w_bl * sr);
            atomicAdd(&trr[tp.i_br], tp.w_br * sr);
        }
    };
    for (int s = j * chunk + threadIdx.x; s < s_end; s += blockDim.x) {
        Ray r = make_ray<HotMath>(px, py, pzo, x2[s], y2[s], pz);
#ifdef SDIRT_SPEC_HEADER
        trace_spec<false, HotMath>(spec_tw, r, conv_mask ? lds_mask : nullptr);
#else
        trace_ray<true, HotMath>(lens, 0, K, kernarg_at(kTripsAt + 64 * w), r, conv_mask ? lds_mask : nullptr);
#endif
        propagate_to<HotMath>(r, zs);
        splat(r.ox, r.oy, r.dx, r.dz, r.ra);
    }
    __syncthreads();

    float* Lg = lout + ((int64_t)n * W + w) * tile;
    float* Rg = HAVE_R ? rout + ((int64_t)n * W + w) * tile : nullptr;
    if (nsplit == 1) {
        if (flags & SDIRT_PSF_NORMALIZE) {
            float mx = -INFINITY;
            for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, tl_[i]);
            const auto div_l = UDiv<HotMath>::make(block_max(mx, red) + 1e-6f);
            auto div_r = div_l;
            if (HAVE_R) {
                mx = -INFINITY;
                for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, trr[i]);
                div_r = UDiv<HotMath>::make(block_max(mx, red) + 1e-6f);
            }
            for (int i = threadIdx.x; i < tile; i += blockDim.x) {
                Lg[i] = div_l(tl_[i]);
                if (HAVE_R) Rg[i] = div_r(trr[i]);
            }
        } else {
            for (int i = threadIdx.x; i < tile; i += blockDim.x) {
                Lg[i] = tl_[i];
                if (HAVE_R) Rg[i] = trr[i];
            }
        }
    } else {
        for (int i = threadIdx.x; i < tile; i += blockDim.x) {
            const float a = tl_[i];
            if (a != 0.0f) atomicAdd(&Lg[i], a);
            if (HAVE_R) {
                const float b = trr[i];
                if (b != 0.0f) atomicAdd(&Rg[i], b);
            }
        }
    }
    if (conv_mask && (int)threadIdx.x < K && lds_mask[threadIdx.x])
        atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
#ifdef SDIRT_DBG_CLOCK
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {      // a workgroup from the middle of the launch
        g_dbg_clock[0] = __builtin_readcyclecounter() - dbg_t0;
        g_dbg_clock[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
#endif
}

#ifdef SDIRT_DBG_CLOCK
extern "C" int sdirt_debug_clock(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_clock), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -3;
}
#endif

// ---------------------------------------------------------------------------
// per-pixel PSF convolution (render_psf.py:76-188)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float round_half(float v) { return (float)(_Float16)v; }

// The fp16 arithmetic of the reference's _fast renderer (x.half() * psf.half(), summed in fp32)
// on packed registers: (wl, wr) rounded to fp16 as a pair, one v_pk_mul_f16 per image value --
// the fp16 product of two fp16 numbers IS round_half(float(v) * float(w)): their exact product
// has 22 significant bits and fits fp32 -- and the two widening accumulations.
typedef _Float16 hpair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hpair half_pair(float lo, float hi) { return hpair{(_Float16)lo, (_Float16)hi}; }
// acc += float(low / high half of the packed pair p), one v_fma_mix_f32 each (the compiler
// splits fma(x, 1, acc) into a conversion and an addition)
__device__ __forceinline__ void acc_halves(hpair p, float& accl, float& accr)
{
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(accl) : "v"(p));
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(accr) : "v"(p));
}
__device__ __forceinline__ void mul_acc_half(float v, hpair w, float& accl, float& accr)
{
    const _Float16 vh = (_Float16)v;
    acc_halves(hpair{vh, vh} * w, accl, accr);
}
// the same with the image value already rounded to fp16 (the pipelined kernel keeps its patch so)
__device__ __forceinline__ void mul_acc_half(_Float16 vh, hpair w, float& accl, float& accr)
{
    acc_halves(hpair{vh, vh} * w, accl, accr);
}

// One WAVE per output pixel: the 64 lanes stride over the 2*ks*ks kernel taps of that
// pixel, so the per-pixel PSFs -- the only large operand, 2*ks*ks*4 B per pixel, read exactly
// once -- stream in as fully coalesced 256-B segments.  The image (a few MB) is gathered
// through L1/L2 with replicate padding (clamped coordinates) and the flipped-tap index of
// render_psf.py:138.  Each lane keeps C partial sums for L and for R; a butterfly of wave
// shuffles reduces them.  A workgroup of 4 waves walks 4 consecutive pixels at a time.
template <int C, bool HALF>
__global__ void __launch_bounds__(kBlock)
k_local_psf_render(const float* __restrict__ img, const float* __restrict__ psf, int B, int H, int W,
                   int ks, float* __restrict__ outl, float* __restrict__ outr)
{
    const int64_t HW = (int64_t)H * W;
    const int64_t P = (int64_t)B * HW;
    const int lane = threadIdx.x & 63;
    const int pad = (ks - 1) / 2, kk = ks * ks;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int rows_per_iter = ks <= 64 ? 64 / ks : 1;
    const int lane_row = ks <= 64 ? lane / ks : 0;
    const int lane_col = ks <= 64 ? lane - lane_row * ks : lane;
    for (int64_t p = wave0; p < P; p += nwaves) {
        const int b = (int)(p / HW);
        const int64_t q = p - (int64_t)b * HW;
        const int y = (int)(q / W), x = (int)(q - (int64_t)y * W);
        const float* kl = psf + p * 2 * kk;
        const float* kr = kl + kk;
        float accl[C], accr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
        // lanes tile the kernel as (rows_per_iter x ks): no integer division inside the loop,
        // consecutive lanes read consecutive taps (and consecutive image columns)
        for (int i0 = 0; i0 < ks; i0 += rows_per_iter) {
            for (int j0 = 0; j0 < ks; j0 += 64) {
                const int fi = i0 + lane_row, fj = j0 + lane_col;
                if (lane_row < rows_per_iter && fi < ks && fj < ks) {
                    const int f = fi * ks + fj;
                    // stored tap f multiplies the neighbour at the FLIPPED offset (render_psf.py:138)
                    const int yy = min(max(y + (ks - 1 - fi) - pad, 0), H - 1);
                    const int xx = min(max(x + (ks - 1 - fj) - pad, 0), W - 1);
                    const float wl = kl[f], wr = kr[f];
                    const hpair wpair = half_pair(wl, wr);
                    const float* px = img + ((int64_t)b * C * H + yy) * W + xx;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float v = px[(int64_t)c * HW];
                        if (HALF) {
                            mul_acc_half(v, wpair, accl[c], accr[c]);
                        } else {
                            accl[c] += v * wl;
                            accr[c] += v * wr;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float a = accl[c], r = accr[c];
            for (int off = 32; off > 0; off >>= 1) {
                a += __shfl_xor(a, off);
                r += __shfl_xor(r, off);
            }
            if (lane == 0) {
                const int64_t o = ((int64_t)(b * C + c) * H + y) * W + x;
                outl[o] = HALF ? round_half(a) : a;
                outr[o] = HALF ? round_half(r) : r;
            }
        }
    }
}

// Sum over the 64 lanes of a wave with DPP row operations (VALU only: no LDS traffic, no
// address registers); the total is returned in every lane.
__device__ __forceinline__ float wave_sum(float v)
{
#define SDIRT_DPP_ADD(CTRL, ROWS)                                                               \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, false))
    SDIRT_DPP_ADD(0xB1, 0xF);     // quad_perm [1,0,3,2]
    SDIRT_DPP_ADD(0x4E, 0xF);     // quad_perm [2,3,0,1]
    SDIRT_DPP_ADD(0x141, 0xF);    // row_half_mirror
    SDIRT_DPP_ADD(0x140, 0xF);    // row_mirror: every lane of a 16-lane row holds the row sum
    SDIRT_DPP_ADD(0x142, 0xA);    // row_bcast:15 -> rows 1 and 3 add the previous row
    SDIRT_DPP_ADD(0x143, 0xC);    // row_bcast:31 -> rows 2 and 3 add rows 0+1
#undef SDIRT_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// LDS-tiled renderer for any kernel size up to 64 and 1 / 3 / 4 channels: a workgroup streams the
// [L | R] kernels of PIX consecutive pixels of one image row (one contiguous run of PIX*2*ks*ks
// floats) into LDS with 16-byte loads -- every PSF byte is read from HBM exactly once, at full
// coalescing width -- then each wave convolves PIX/4 of those pixels reading its weights from LDS.
// blockIdx.y is an image row (b * H + y), blockIdx.x strides over that row's pixel groups.
// Round 1's kernel walked a flat pixel index and paid a 64-bit division per pixel to recover
// (b, y, x) -- a software sequence of ~150 scalar and vector instructions, a third of its
// instruction stream.  Here (b, y) cost one 32-bit division per workgroup, the tap geometry of a
// lane (its row / column inside the kernel, its flipped offsets, its LDS index) is computed once,
// and the per-tap work is: one clamp of the row coordinate, one address, C image gathers, two LDS
// reads, the fp16 arithmetic.  The group's [L | R] kernels are one contiguous run of the PSF
// tensor; it is copied with 16-byte loads whatever its alignment (the LDS image is shifted by the
// run's misalignment so that source and destination stay congruent modulo 16 bytes).
template <int C, bool HALF, int PIX, int KS>
__global__ void __launch_bounds__(kBlock)
k_local_psf_render_rows(const float* __restrict__ img, const float* __restrict__ psf, int H, int W,
                        int ks_rt, float* __restrict__ outl, float* __restrict__ outr)
{
    const int ks = KS > 0 ? KS : ks_rt;                              // ks <= 64 on this path
    extern __shared__ __attribute__((aligned(16))) float wts[];     // 4 + [PIX][2][ks*ks]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pad = (ks - 1) / 2, kk = ks * ks;
    const int row = blockIdx.y;
    const int b = row / H, y = row - b * H;
    const int HW = H * W;
    const float* __restrict__ img_b = img + (int64_t)b * C * HW;
    // this lane's tap: lanes tile the kernel as (rows_per_iter x ks)
    const int rows_per_iter = 64 / ks;
    const int lane_row = lane / ks, lane_col = lane - lane_row * ks;
    const bool lane_on = lane_row < rows_per_iter;
    const int dx = (ks - 1 - lane_col) - pad;                        // flipped tap -> neighbour offset (render_psf.py:138)
    const int dy0 = (ks - 1 - lane_row) - pad;
    const int f0 = lane_row * ks + lane_col;
    const int groups = (W + PIX - 1) / PIX;
    typedef float fl4 __attribute__((ext_vector_type(4)));
    for (int gx = blockIdx.x; gx < groups; gx += gridDim.x) {
        const int x0 = gx * PIX;
        const int npix = min(PIX, W - x0);
        const int64_t first = ((int64_t)row * W + x0) * 2 * kk;       // first float of the run
        const int nfl = npix * 2 * kk;
        const int sh = (int)(first & 3);                              // misalignment, in floats
        const float* src = psf + first;
        float* dst = wts + sh;
        const int head = min((4 - sh) & 3, nfl);
        const int nf4 = (nfl - head) >> 2;
        const fl4* src4 = reinterpret_cast<const fl4*>(src + head);
        fl4* dst4 = reinterpret_cast<fl4*>(dst + head);
        constexpr int STAGE_U = 8;
        for (int base = threadIdx.x; base < nf4; base += kBlock * STAGE_U) {
            fl4 v[STAGE_U];
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u)
                if (base + u * kBlock < nf4) v[u] = __builtin_nontemporal_load(&src4[base + u * kBlock]);
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u)
                if (base + u * kBlock < nf4) dst4[base + u * kBlock] = v[u];
        }
        if ((int)threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
        for (int i = head + (nf4 << 2) + threadIdx.x; i < nfl; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        for (int q = wave; q < npix; q += kBlock / 64) {
            const int x = x0 + q;
            const int xx = min(max(x + dx, 0), W - 1);
            const float* kl = dst + q * 2 * kk + f0;
            float accl[C], accr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
            if (KS > 0) {
                // compile-time kernel size: ALL image gathers of the pixel are issued before the first
                // one is used (the kernel is bound by the latency of these L2 hits, not by their count)
                constexpr int NI = KS > 0 ? (KS + (64 / (KS > 0 ? KS : 1)) - 1) / (64 / (KS > 0 ? KS : 1)) : 1;
                float v[NI][C];
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int i0 = it * rows_per_iter;
                    const int yy = min(max(y + dy0 - i0, 0), H - 1);
                    const int off = yy * W + xx;
#pragma unroll
                    for (int c = 0; c < C; ++c) v[it][c] = img_b[c * HW + off];
                }
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int i0 = it * rows_per_iter;
                    const bool on = lane_on && i0 + lane_row < ks;
                    const float wl = on ? kl[i0 * ks] : 0.0f, wr = on ? kl[kk + i0 * ks] : 0.0f;
                    const hpair wpair = half_pair(wl, wr);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        if (HALF) {
                            mul_acc_half(v[it][c], wpair, accl[c], accr[c]);
                        } else {
                            accl[c] += v[it][c] * wl;
                            accr[c] += v[it][c] * wr;
                        }
                    }
                }
            } else {
                for (int i0 = 0; i0 < ks; i0 += rows_per_iter) {
                    if (lane_on && i0 + lane_row < ks) {
                        const int yy = min(max(y + dy0 - i0, 0), H - 1);
                        const float wl = kl[i0 * ks], wr = kl[kk + i0 * ks];
                        const hpair wpair = half_pair(wl, wr);
                        const int off = yy * W + xx;
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float vv = img_b[c * HW + off];
                            if (HALF) {
                                mul_acc_half(vv, wpair, accl[c], accr[c]);
                            } else {
                                accl[c] += vv * wl;
                                accr[c] += vv * wr;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float a = wave_sum(accl[c]), rr = wave_sum(accr[c]);
                if (lane == 0) {
                    const int64_t o = ((int64_t)(b * C + c) * H + y) * W + x;
                    outl[o] = HALF ? round_half(a) : a;
                    outr[o] = HALF ? round_half(rr) : rr;
                }
            }
        }
        __syncthreads();
    }
}

// Software-pipelined renderer for a compile-time kernel size (the reference's ks 21): while a
// workgroup convolves the 8 pixels of group g out of LDS, the loads of group g+1 -- its 28 KB of
// [L | R] kernels AND the (KS x (8 + KS - 1)) x C image patch the 8 pixels read, replicate-clamped --
// are in flight into registers.  The compute phase issues NO global loads (vmcnt returns in order:
// a gather issued behind the prefetch would wait for all of it), its image reads are LDS reads at
// immediate offsets, no coordinate clamps.  k_local_psf_render_rows staged, computed, staged, ...
// with the phases of all workgroups of a CU aligned: 0.44 ms against a 0.19 ms read-only stream of
// the same bytes.
template <int C, bool HALF, int KS, int PIX>
__global__ void __launch_bounds__(kBlock)
k_local_psf_render_pipe(const float* __restrict__ img, const float* __restrict__ psf, int H, int W,
                        int64_t total_floats, float* __restrict__ outl, float* __restrict__ outr)
{
    constexpr int kk = KS * KS, pad = (KS - 1) / 2;
    constexpr int PW = PIX + KS - 1;                 // patch width
    constexpr int NI = (kk + 63) / 64;               // wave passes over the kernel's taps
    constexpr int NPSF = PIX * 2 * kk;
    constexpr int NV = (NPSF / 4 + 1 + kBlock - 1) / kBlock;      // fl4 per thread (run + shift slack)
    constexpr int NPATCH = C * KS * PW;
    constexpr int NQ = (NPATCH + kBlock - 1) / kBlock;
    typedef float fl4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wts = lds;                                // [NPSF + 4] (the run, shifted by its misalignment)
    // the image patch [C][KS][PW]: rounded to fp16 once, at staging, for the fp16 arithmetic of the
    // _fast renderer (every element is read by up to 8 x KS taps), fp32 for the fp32 renderer
    typedef typename std::conditional<HALF, _Float16, float>::type PatchT;
    PatchT* patch = reinterpret_cast<PatchT*>(lds + ((NPSF + 4 + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y;
    const int b = row / H, y = row - b * H;
    const int HW = H * W;
    const float* __restrict__ img_b = img + (int64_t)b * C * HW;
    // pass `it` of a wave covers taps f = 64 it + lane of the kernel (consecutive lanes read
    // consecutive weights); tap (fi, fj) multiplies the neighbour at the flipped offset
    // (render_psf.py:138), i.e. patch row KS-1-fi, patch column q + KS-1-fj.  Only the last pass has
    // lanes without a tap (448 - 441 of them at ks 21).
    int ptap[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = min(it * 64 + lane, kk - 1);
        const int fi = f / KS, fj = f - fi * KS;
        ptap[it] = (KS - 1 - fi) * PW + (KS - 1 - fj);
    }
    const int groups = (W + PIX - 1) / PIX;
    const int64_t total4 = total_floats >> 2;

    fl4 pv[NV];
    float pq[NQ];
    int sh_next = 0;
    // patch element e = threadIdx.x + u * kBlock of this thread: its image row never changes (the
    // workgroup stays on one row), its column moves with the group
    int prow[NQ], pcol[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int e = min((int)threadIdx.x + u * kBlock, NPATCH - 1);
        const int c = e / (KS * PW), r = (e - c * KS * PW) / PW;
        pcol[u] = e - c * KS * PW - r * PW - pad;
        prow[u] = c * HW + min(max(y + r - pad, 0), H - 1) * W;
    }
    const fl4* src4 = reinterpret_cast<const fl4*>(psf);
    auto fetch = [&](int gx) {
        const int x0 = gx * PIX;
        const int npix = min(PIX, W - x0);
        const int64_t first = ((int64_t)row * W + x0) * 2 * kk;
        const int sh = (int)(first & 3);
        sh_next = sh;
        const int64_t v0 = (first - sh) >> 2;                       // first 16-byte vector of the run
        const int nv = (sh + npix * 2 * kk + 3) >> 2;
        if (v0 + nv <= total4) {                                    // (always, but for the tensor's last bytes)
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = threadIdx.x + u * kBlock;
                if (u + 1 < NV || i < nv) pv[u] = __builtin_nontemporal_load(&src4[v0 + min(i, nv - 1)]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = threadIdx.x + u * kBlock;
                fl4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                if (i < nv)
                    for (int e = 0; e < 4; ++e)
                        if (((v0 + i) << 2) + e < total_floats) t[e] = psf[((v0 + i) << 2) + e];
                pv[u] = t;
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            pq[u] = img_b[prow[u] + min(max(x0 + pcol[u], 0), W - 1)];
    };
    auto commit = [&](int gx) {
        const int npix = min(PIX, W - gx * PIX);
        const int nv = (sh_next + npix * 2 * kk + 3) >> 2;
        fl4* dst4 = reinterpret_cast<fl4*>(wts);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = threadIdx.x + u * kBlock;
            if (i < nv) dst4[i] = pv[u];
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int e = threadIdx.x + u * kBlock;
            if (e < NPATCH) patch[e] = (PatchT)pq[u];
        }
    };

    int gx = blockIdx.x;
    if (gx >= groups) return;
#ifdef SDIRT_DBG_CLOCK
    const unsigned long long dbg_t0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    fetch(gx);
    commit(gx);
    int sh = sh_next;
    __syncthreads();
    for (; gx < groups; gx += gridDim.x) {
        const int gn = gx + gridDim.x;
#ifndef SDIRT_ABL_NOFETCH
        if (gn < groups) fetch(gn);                  // in flight during the convolution below
#endif
        const int x0 = gx * PIX;
#ifdef SDIRT_ABL_NOCOMPUTE
        const int npix = 0;
#else
        const int npix = min(PIX, W - x0);
#endif
        for (int q = wave; q < npix; q += kBlock / 64) {
            // every lane reads unconditionally (a predicated read costs a branch and a wait of its
            // own); the tap-less lanes of the last pass read in-bounds elements and are zeroed
            const float* kl = wts + sh + q * 2 * kk + lane;
            float accl[C], accr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
            float rwl[NI], rwr[NI];
            PatchT rv[NI][C];
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const bool in = it * 64 + 63 < kk;                   // compile-time: whole pass inside the kernel
                const int f = in ? it * 64 : min(it * 64, kk - 1 - lane);
                rwl[it] = kl[f];
                rwr[it] = kl[kk + f];
                const PatchT* pp = patch + ptap[it] + q;
#pragma unroll
                for (int c = 0; c < C; ++c) rv[it][c] = pp[c * KS * PW];
            }
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const bool on = it * 64 + 63 < kk || it * 64 + lane < kk;
                const float wl = on ? rwl[it] : 0.0f, wr = on ? rwr[it] : 0.0f;
                const hpair wpair = half_pair(wl, wr);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const PatchT v = on ? rv[it][c] : (PatchT)0.0f;
                    if (HALF) {
                        mul_acc_half(v, wpair, accl[c], accr[c]);
                    } else {
                        accl[c] += v * wl;
                        accr[c] += v * wr;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float a = wave_sum(accl[c]), rr = wave_sum(accr[c]);
                if (lane == 0) {
                    const int64_t o = ((int64_t)(b * C + c) * H + y) * W + x0 + q;
                    outl[o] = HALF ? round_half(a) : a;
                    outr[o] = HALF ? round_half(rr) : rr;
                }
            }
        }
        __syncthreads();                             // everyone is done reading this group's LDS image
        if (gn < groups) {
            commit(gn);
            sh = sh_next;
        }
        __syncthreads();
    }
#ifdef SDIRT_DBG_CLOCK
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2) {
        g_dbg_clock[2] = __builtin_readcyclecounter() - dbg_t0;
        g_dbg_clock[3] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
#endif
}

// Six wave sums for the price of two.  A butterfly level that pairs lanes l and l^16 / l^32 is a
// gfx950 half-exchange (v_permlane16_swap / v_permlane32_swap: the odd rows / the upper half of
// one register trade places with the even rows / the lower half of another) plus ONE addition
// for TWO vectors, whose sums end up in different rows of the result: after both levels
// `q` holds, per 16-lane row, the partial sums of (a0, a1, a2, b0) and `s` those of (b1, b2, b1, b2);
// four DPP additions inside the rows finish both.  19 vector instructions for six sums, against
// 6 x (6 DPP additions + v_readlane).  Row r of q / s: every lane holds the total.
__device__ __forceinline__ float swap16_add(float a, float b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows: a(0+1) b(0+1) a(2+3) b(2+3)
}
__device__ __forceinline__ float swap32_add(float a, float b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // lanes 0-31: a(lo+hi), 32-63: b(lo+hi)
}
__device__ __forceinline__ float row_sum(float v)
{
#define SDIRT_DPP_ADD(CTRL)                                                                     \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false))
    SDIRT_DPP_ADD(0xB1);      // quad_perm [1,0,3,2]
    SDIRT_DPP_ADD(0x4E);      // quad_perm [2,3,0,1]
    SDIRT_DPP_ADD(0x141);     // row_half_mirror
    SDIRT_DPP_ADD(0x140);     // row_mirror
#undef SDIRT_DPP_ADD
    return v;
}
__device__ __forceinline__ void wave_sum6(const float (&a)[3], const float (&b)[3], float& q, float& s)
{
    const float p01 = swap16_add(a[0], a[1]), p23 = swap16_add(a[2], b[0]), p45 = swap16_add(b[1], b[2]);
    q = row_sum(swap32_add(p01, p23));          // rows: a0 a1 a2 b0
    s = row_sum(swap32_add(p45, p45));          // rows: b1 b2 b1 b2
}

// One wave per pixel, weights straight from HBM into registers (ks 21, RGB).
// k_local_psf_render_pipe moved every weight through LDS (16-byte loads, a write, a read per
// weight, two barriers per 8 pixels) and spent 165 instructions per pixel and wave.  Here lane l
// loads the weights of ITS taps (f = 64 it + l: consecutive lanes, consecutive floats -- each load
// instruction of a wave is one contiguous 256 bytes of the pixel's kernel) one pixel ahead of the
// one being convolved; LDS holds only the image patch of the workgroup's CHUNK-pixel stretch of
// the row ([KS][CHUNK + KS - 1] positions x 4 channel slots: one 8- or 16-byte read per tap gives
// all channels), staged once: one barrier per workgroup, none around the weights.  The six sums
// of a pixel are reduced together (wave_sum6) and stored by 4 + 2 lanes in two instructions.
template <int C, bool HALF, int KS, int CHUNK>
__global__ void __launch_bounds__(kBlock)
#ifdef SDIRT_RENDER_WAVES
__attribute__((amdgpu_waves_per_eu(SDIRT_RENDER_WAVES, SDIRT_RENDER_WAVES)))
#endif
k_local_psf_render_wave(const float* __restrict__ img, const float* __restrict__ psf, int H, int W,
                        float* __restrict__ outl, float* __restrict__ outr)
{
    static_assert(C == 3, "row layout of wave_sum6");
    constexpr int kk = KS * KS, pad = (KS - 1) / 2;
    constexpr int PW = CHUNK + KS - 1;               // patch width
    constexpr int NI = (kk + 63) / 64;               // taps per lane
    constexpr int NPOS = KS * PW;                    // patch positions
    constexpr int NQ = (NPOS + kBlock - 1) / kBlock;
    constexpr int NW = kBlock / 64, PPW = CHUNK / NW;   // waves, pixels per wave
    static_assert(PPW % 2 == 0, "the pixel loop is unrolled by two");
    typedef typename std::conditional<HALF, _Float16, float>::type PatchT;
    typedef PatchT pvec __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    pvec* patch = reinterpret_cast<pvec*>(lds_raw);  // [NPOS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y;
    const int b = row / H, y = row - b * H;
    const int HW = H * W;
    const int x0 = blockIdx.x * CHUNK;
    const float* __restrict__ wrow = psf + (int64_t)row * W * 2 * kk;      // this image row's kernels
    const int ftail = min((NI - 1) * 64 + lane, kk - 1);
    const bool tail_on = (NI - 1) * 64 + lane < kk;
    auto load_w = [&](int x, float (&l)[NI], float (&r)[NI]) {
        const float* __restrict__ k0 = wrow + (int64_t)min(x, W - 1) * 2 * kk + lane;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int f = it + 1 < NI ? it * 64 : ftail - lane;
#ifdef SDIRT_ABL_NOFETCH
            l[it] = __int_as_float(0x3c000000 + f + x); r[it] = __int_as_float(0x3c100000 + f + x);   // ablation: no loads
#else
#ifdef SDIRT_RENDER_PLAIN_LOADS
            l[it] = k0[f];
            r[it] = k0[kk + f];
#else
            l[it] = __builtin_nontemporal_load(k0 + f);
            r[it] = __builtin_nontemporal_load(k0 + kk + f);
#endif
#endif
        }
    };
    float wa[NI], ra[NI], wb[NI], rb[NI];
    load_w(x0 + wave, wa, ra);                       // in flight while the patch is staged
    {
        const float* __restrict__ img_b = img + (int64_t)b * C * HW;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int e = threadIdx.x + u * kBlock;
            if (e < NPOS) {
                const int r = e / PW, col = e - r * PW - pad;
                const int o = min(max(y + r - pad, 0), H - 1) * W + min(max(x0 + col, 0), W - 1);
                pvec v;
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = (PatchT)img_b[c * HW + o];
                v[3] = (PatchT)0.0f;
                patch[e] = v;
            }
        }
    }
    // tap f = 64 it + l of the kernel multiplies the neighbour at the flipped offset
    // (render_psf.py:138): patch row KS-1-fi, patch column q + KS-1-fj.  The lanes past the last
    // tap (448 - 441 at ks 21) read tap kk-1 and are zeroed.
    int ptap[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = min(it * 64 + lane, kk - 1);
        const int fi = f / KS, fj = f - fi * KS;
        ptap[it] = (KS - 1 - fi) * PW + (KS - 1 - fj);
    }
    // this lane's output slot: rows 0..2 of the reduced vector q are the L channels, row 3 is R
    // channel 0; rows 0, 1 of s are R channels 1, 2
    const int r16 = lane >> 4;
    float* __restrict__ oq = (r16 < 3 ? outl + ((int64_t)(b * C + r16) * H + y) * W
                                      : outr + ((int64_t)(b * C) * H + y) * W);
    float* __restrict__ os = outr + ((int64_t)(b * C + 1 + (r16 & 1)) * H + y) * W;
    const bool store_q = (lane & 15) == 0, store_s = (lane & 47) == 0;
    __syncthreads();

    auto pixel = [&](int x, const float (&wl)[NI], const float (&wr)[NI]) {
        if (x >= W) return;
#ifdef SDIRT_ABL_NOCOMPUTE
        {   // ablation: consume the weights with one addition each, store one value
            float t = 0.0f;
#pragma unroll
            for (int it = 0; it < NI; ++it) t += wl[it] + wr[it];
            if (t == 12345.0f) oq[x] = t;
            return;
        }
#endif
        const pvec* pp = patch + (x - x0);
        float accl[C], accr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const bool on = it + 1 < NI || tail_on;
            const float a = on ? wl[it] : 0.0f, bq = on ? wr[it] : 0.0f;
            const pvec v = pp[ptap[it]];
            if (HALF) {
                const hpair wpair = half_pair(a, bq);
#pragma unroll
                for (int c = 0; c < C; ++c) mul_acc_half((_Float16)v[c], wpair, accl[c], accr[c]);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    accl[c] += (float)v[c] * a;
                    accr[c] += (float)v[c] * bq;
                }
            }
        }
        float q, s2;
        wave_sum6(accl, accr, q, s2);
        if (HALF) { q = round_half(q); s2 = round_half(s2); }
        if (store_q) oq[x] = q;
        if (store_s) os[x] = s2;
    };
#pragma unroll 1
    for (int j = 0; j < PPW; j += 2) {
        const int x = x0 + wave + j * NW;
        load_w(x + NW, wb, rb);
        pixel(x, wa, ra);
        if (j + 2 < PPW) load_w(x + 2 * NW, wa, ra);
        pixel(x + NW, wb, rb);
    }
}

// PSFNet.pred (psfnet.py:317-336) + local_psf_render_fast (render_psf.py:120-155) in one pass
// over the network's raw fp16 outputs: raw_l = net(x, y, z), raw_r = net(-x, y, z), both
// [P, ks*ks].  Per pixel: L taps = raw_l / (sum(raw_l) + 1e-9), R taps = fliplr(raw_r) /
// (sum(raw_r) + 1e-9), then the per-pixel convolution with the fp16 arithmetic of the _fast
// renderer.  The stacked / flipped / normalised [P,2,ks,ks] tensor the reference materialises
// (and re-reads twice) never exists: each raw value is read from HBM once, as fp16.
// A zero-sum kernel renders 0 (the reference's fp16 division would give NaN there).
template <int C, int PIX, int KS>
__global__ void __launch_bounds__(kBlock)
k_psfnet_render(const float* __restrict__ img, const _Float16* __restrict__ raw_l,
                const _Float16* __restrict__ raw_r, int B, int H, int W, int ks_rt,
                float* __restrict__ outl, float* __restrict__ outr)
{
    const int ks = KS > 0 ? KS : ks_rt;
    extern __shared__ __attribute__((aligned(16))) _Float16 wh[];   // [2][PIX][ks*ks]
    const int64_t HW = (int64_t)H * W;
    const int64_t P = (int64_t)B * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pad = (ks - 1) / 2, kk = ks * ks;
    const int rows_per_iter = ks <= 64 ? 64 / ks : 1;
    const int lane_row = ks <= 64 ? lane / ks : 0;
    const int lane_col = ks <= 64 ? lane - lane_row * ks : lane;
    const int64_t ngroups = (P + PIX - 1) / PIX;
    typedef float fl4 __attribute__((ext_vector_type(4)));
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t p0 = g * PIX;
        const int npix = (int)min((int64_t)PIX, P - p0);
        const int nh = npix * kk;                                   // halves per side
        const int nv = nh >> 3;                                     // 16-byte vectors per side
        // PIX is a multiple of 8, so both runs start 16-byte aligned
        const fl4* sl4 = reinterpret_cast<const fl4*>(raw_l + p0 * kk);
        const fl4* sr4 = reinterpret_cast<const fl4*>(raw_r + p0 * kk);
        fl4* dl4 = reinterpret_cast<fl4*>(wh);
        fl4* dr4 = reinterpret_cast<fl4*>(wh + PIX * kk);
        constexpr int STAGE_U = 4;
        for (int base = threadIdx.x; base < nv; base += kBlock * STAGE_U) {
            fl4 a[STAGE_U], b[STAGE_U];
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u)
                if (base + u * kBlock < nv) {
                    a[u] = __builtin_nontemporal_load(&sl4[base + u * kBlock]);
                    b[u] = __builtin_nontemporal_load(&sr4[base + u * kBlock]);
                }
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u)
                if (base + u * kBlock < nv) { dl4[base + u * kBlock] = a[u]; dr4[base + u * kBlock] = b[u]; }
        }
        for (int i = (nv << 3) + threadIdx.x; i < nh; i += blockDim.x) {
            wh[i] = raw_l[p0 * kk + i];
            wh[PIX * kk + i] = raw_r[p0 * kk + i];
        }
        __syncthreads();
        for (int q = wave; q < npix; q += kBlock / 64) {
            const int64_t p = p0 + q;
            const int b = (int)(p / HW);
            const int64_t r = p - (int64_t)b * HW;
            const int y = (int)(r / W), x = (int)(r - (int64_t)y * W);
            const _Float16* kl = wh + q * kk;
            const _Float16* kr = wh + PIX * kk + q * kk;
            float sl = 0.0f, sr = 0.0f;
            for (int f = lane; f < kk; f += 64) { sl += (float)kl[f]; sr += (float)kr[f]; }
            const float inv_l = 1.0f / (round_half(wave_sum(sl)) + 1e-9f);
            const float inv_r = 1.0f / (round_half(wave_sum(sr)) + 1e-9f);
            float accl[C], accr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
#pragma unroll KS > 0 ? 8 : 1
            for (int i0 = 0; i0 < ks; i0 += rows_per_iter) {
                for (int j0 = 0; j0 < ks; j0 += 64) {
                    const int fi = i0 + lane_row, fj = j0 + lane_col;
                    if (lane_row < rows_per_iter && fi < ks && fj < ks) {
                        const int yy = min(max(y + (ks - 1 - fi) - pad, 0), H - 1);
                        const int xx = min(max(x + (ks - 1 - fj) - pad, 0), W - 1);
                        const hpair wpair = half_pair((float)kl[fi * ks + fj] * inv_l,
                                                      (float)kr[fi * ks + (ks - 1 - fj)] * inv_r);
                        const float* px = img + ((int64_t)b * C * H + yy) * W + xx;
#pragma unroll
                        for (int c = 0; c < C; ++c) mul_acc_half(px[(int64_t)c * HW], wpair, accl[c], accr[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float a = wave_sum(accl[c]), rr = wave_sum(accr[c]);
                if (lane == 0) {
                    const int64_t o = ((int64_t)(b * C + c) * H + y) * W + x;
                    outl[o] = round_half(a);
                    outr[o] = round_half(rr);
                }
            }
        }
        __syncthreads();
    }
}

// k_psfnet_render with the structure of k_local_psf_render_pipe (compile-time kernel size, one image
// row per workgroup row, 8-pixel groups): the two fp16 runs of group g+1 and its image patch are in
// flight into registers while group g is convolved out of LDS; image patch kept in fp16; taps
// mapped linearly onto lanes (f = 64 it + lane); accumulation with v_fma_mix_f32.  Same arithmetic,
// same results as k_psfnet_render.
template <int C, int KS>
__global__ void __launch_bounds__(kBlock)
k_psfnet_render_pipe(const float* __restrict__ img, const _Float16* __restrict__ raw_l,
                     const _Float16* __restrict__ raw_r, int H, int W, int64_t total_halves,
                     float* __restrict__ outl, float* __restrict__ outr)
{
    constexpr int PIX = 8, kk = KS * KS, pad = (KS - 1) / 2;
    constexpr int PW = PIX + KS - 1;
    constexpr int NI = (kk + 63) / 64;
    constexpr int NH = PIX * kk;                                   // halves per side and group
    constexpr int SIDE = (NH + 8 + 7) & ~7;                        // LDS halves per side (+ shift slack)
    constexpr int NV = (NH / 8 + 1 + kBlock - 1) / kBlock;         // 16-byte vectors per thread and side
    constexpr int NPATCH = C * KS * PW;
    constexpr int NQ = (NPATCH + kBlock - 1) / kBlock;
    typedef float fl4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) _Float16 ldh[];
    _Float16* wl_ = ldh;                                           // [SIDE]
    _Float16* wr_ = ldh + SIDE;                                    // [SIDE]
    _Float16* patch = ldh + 2 * SIDE;                              // [C][KS][PW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y;
    const int b = row / H, y = row - b * H;
    const int HW = H * W;
    const float* __restrict__ img_b = img + (int64_t)b * C * HW;
    int ptap[NI], fflip[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = min(it * 64 + lane, kk - 1);
        const int fi = f / KS, fj = f - fi * KS;
        ptap[it] = (KS - 1 - fi) * PW + (KS - 1 - fj);
        fflip[it] = fi * KS + (KS - 1 - fj);                        // fliplr of the right kernel (psfnet.py:330)
    }
    const int groups = (W + PIX - 1) / PIX;
    const int64_t total8 = total_halves >> 3;

    fl4 pl[NV], pr[NV];
    float pq[NQ];
    int prow[NQ], pcol[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int e = min((int)threadIdx.x + u * kBlock, NPATCH - 1);
        const int c = e / (KS * PW), r = (e - c * KS * PW) / PW;
        pcol[u] = e - c * KS * PW - r * PW - pad;
        prow[u] = c * HW + min(max(y + r - pad, 0), H - 1) * W;
    }
    int sh_next = 0;
    const fl4* sl4 = reinterpret_cast<const fl4*>(raw_l);
    const fl4* sr4 = reinterpret_cast<const fl4*>(raw_r);
    auto fetch = [&](int gx) {
        const int x0 = gx * PIX;
        const int npix = min(PIX, W - x0);
        const int64_t first = ((int64_t)row * W + x0) * kk;           // first half of both runs
        const int sh = (int)(first & 7);
        sh_next = sh;
        const int64_t v0 = (first - sh) >> 3;
        const int nv = (sh + npix * kk + 7) >> 3;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = min((int)threadIdx.x + u * kBlock, nv - 1);
            if (v0 + i < total8) {
                pl[u] = __builtin_nontemporal_load(&sl4[v0 + i]);
                pr[u] = __builtin_nontemporal_load(&sr4[v0 + i]);
            } else {                                                // the tensors' last, partial vector
                union { fl4 v; _Float16 h[8]; } a, c;
                for (int e = 0; e < 8; ++e) {
                    const int64_t k = ((v0 + i) << 3) + e;
                    a.h[e] = k < total_halves ? raw_l[k] : (_Float16)0.0f;
                    c.h[e] = k < total_halves ? raw_r[k] : (_Float16)0.0f;
                }
                pl[u] = a.v; pr[u] = c.v;
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            pq[u] = img_b[prow[u] + min(max(x0 + pcol[u], 0), W - 1)];
    };
    auto commit = [&](int gx) {
        const int npix = min(PIX, W - gx * PIX);
        const int nv = (sh_next + npix * kk + 7) >> 3;
        fl4* dl4 = reinterpret_cast<fl4*>(wl_);
        fl4* dr4 = reinterpret_cast<fl4*>(wr_);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = threadIdx.x + u * kBlock;
            if (i < nv) { dl4[i] = pl[u]; dr4[i] = pr[u]; }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int e = threadIdx.x + u * kBlock;
            if (e < NPATCH) patch[e] = (_Float16)pq[u];
        }
    };

    int gx = blockIdx.x;
    if (gx >= groups) return;
    fetch(gx);
    commit(gx);
    int sh = sh_next;
    __syncthreads();
    for (; gx < groups; gx += gridDim.x) {
        const int gn = gx + gridDim.x;
        if (gn < groups) fetch(gn);
        const int x0 = gx * PIX;
        const int npix = min(PIX, W - x0);
        for (int q = wave; q < npix; q += kBlock / 64) {
            const _Float16* kl = wl_ + sh + q * kk;
            const _Float16* kr = wr_ + sh + q * kk;
            float rl[NI], rrf[NI], sl = 0.0f, sr = 0.0f;
            _Float16 rv[NI][C];
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const bool on = it * 64 + 63 < kk || it * 64 + lane < kk;
                const int f = min(it * 64 + lane, kk - 1);
                const float a = (float)kl[f], c = (float)kr[f];
                rl[it] = on ? a : 0.0f;
                sl += on ? a : 0.0f;
                sr += on ? c : 0.0f;
                rrf[it] = on ? (float)kr[fflip[it]] : 0.0f;
                const _Float16* pp = patch + ptap[it] + q;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) rv[it][ch] = pp[ch * KS * PW];
            }
            const float inv_l = 1.0f / (round_half(wave_sum(sl)) + 1e-9f);
            const float inv_r = 1.0f / (round_half(wave_sum(sr)) + 1e-9f);
            float accl[C], accr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const hpair wpair = half_pair(rl[it] * inv_l, rrf[it] * inv_r);
#pragma unroll
                for (int c = 0; c < C; ++c) mul_acc_half(rv[it][c], wpair, accl[c], accr[c]);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float a = wave_sum(accl[c]), rr = wave_sum(accr[c]);
                if (lane == 0) {
                    const int64_t o = ((int64_t)(b * C + c) * H + y) * W + x0 + q;
                    outl[o] = round_half(a);
                    outr[o] = round_half(rr);
                }
            }
        }
        __syncthreads();
        if (gn < groups) {
            commit(gn);
            sh = sh_next;
        }
        __syncthreads();
    }
}

// k_psfnet_render with the structure of k_local_psf_render_wave: one wave per pixel, lane l loads the
// raw network outputs of ITS taps (left: f = 64 it + l; right: the fliplr'ed tap, psfnet.py:330 --
// a permutation of the taps, so its values also make up the right kernel's sum) straight from HBM
// one pixel ahead, the image patch of the workgroup's CHUNK pixels sits in LDS with the three
// channels of a position in one 8-byte slot, both normalising sums are reduced together and the
// six outputs with wave_sum6.  Results equal k_psfnet_render's up to the order of the fp32 sums.
template <int C, int KS, int CHUNK>
__global__ void __launch_bounds__(kBlock)
#ifdef SDIRT_RENDER_WAVES
__attribute__((amdgpu_waves_per_eu(SDIRT_RENDER_WAVES, SDIRT_RENDER_WAVES)))
#endif
k_psfnet_render_wave(const float* __restrict__ img, const _Float16* __restrict__ raw_l,
                     const _Float16* __restrict__ raw_r, int H, int W,
                     float* __restrict__ outl, float* __restrict__ outr)
{
    static_assert(C == 3, "row layout of wave_sum6");
    constexpr int kk = KS * KS, pad = (KS - 1) / 2;
    constexpr int PW = CHUNK + KS - 1;
    constexpr int NI = (kk + 63) / 64;
    constexpr int NPOS = KS * PW;
    constexpr int NQ = (NPOS + kBlock - 1) / kBlock;
    constexpr int NW = kBlock / 64, PPW = CHUNK / NW;
    static_assert(PPW % 2 == 0, "the pixel loop is unrolled by two");
    typedef _Float16 pvec __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    pvec* patch = reinterpret_cast<pvec*>(lds_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y;
    const int b = row / H, y = row - b * H;
    const int HW = H * W;
    const int x0 = blockIdx.x * CHUNK;
    const _Float16* __restrict__ lrow = raw_l + (int64_t)row * W * kk;
    const _Float16* __restrict__ rrow = raw_r + (int64_t)row * W * kk;
    int ptap[NI], fr[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = min(it * 64 + lane, kk - 1);
        const int fi = f / KS, fj = f - fi * KS;
        ptap[it] = (KS - 1 - fi) * PW + (KS - 1 - fj);
        fr[it] = fi * KS + (KS - 1 - fj);
    }
    const int ftail = min((NI - 1) * 64 + lane, kk - 1);
    const bool tail_on = (NI - 1) * 64 + lane < kk;
    auto load_w = [&](int x, _Float16 (&l)[NI], _Float16 (&r)[NI]) {
        const int k0 = min(x, W - 1) * kk;             // a row's runs fit 32-bit offsets (checked by the host)
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            l[it] = __builtin_nontemporal_load(lrow + (k0 + (it + 1 < NI ? it * 64 + lane : ftail)));
            r[it] = __builtin_nontemporal_load(rrow + (k0 + fr[it]));
        }
    };
    _Float16 wa[NI], ra[NI], wb[NI], rb[NI];
    load_w(x0 + wave, wa, ra);
    {
        const float* __restrict__ img_b = img + (int64_t)b * C * HW;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int e = threadIdx.x + u * kBlock;
            if (e < NPOS) {
                const int r = e / PW, col = e - r * PW - pad;
                const int o = min(max(y + r - pad, 0), H - 1) * W + min(max(x0 + col, 0), W - 1);
                pvec v;
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = (_Float16)img_b[c * HW + o];
                v[3] = (_Float16)0.0f;
                patch[e] = v;
            }
        }
    }
    const int r16 = lane >> 4;
    float* __restrict__ oq = (r16 < 3 ? outl + ((int64_t)(b * C + r16) * H + y) * W
                                      : outr + ((int64_t)(b * C) * H + y) * W);
    float* __restrict__ os = outr + ((int64_t)(b * C + 1 + (r16 & 1)) * H + y) * W;
    const bool store_q = (lane & 15) == 0, store_s = (lane & 47) == 0;
    __syncthreads();

    auto pixel = [&](int x, const _Float16 (&hl)[NI], const _Float16 (&hr)[NI]) {
        if (x >= W) return;
        const pvec* pp = patch + (x - x0);
        float wl[NI], wr[NI], sl = 0.0f, sr = 0.0f;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const bool on = it + 1 < NI || tail_on;
            wl[it] = on ? (float)hl[it] : 0.0f;
            wr[it] = on ? (float)hr[it] : 0.0f;
            sl += wl[it];
            sr += wr[it];
        }
        // both sums at once: even rows of t end up with sum(sl), odd rows with sum(sr)
        const float t = row_sum(swap32_add(swap16_add(sl, sr), swap16_add(sl, sr)));
        const float tot_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
        const float tot_r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16));
        // psf / (psf.sum() + 1e-9) in half precision (psfnet.py:333): the sum rounded to fp16
        const float inv_l = sdirt::Lean::div(1.0f, round_half(tot_l) + 1e-9f);
        const float inv_r = sdirt::Lean::div(1.0f, round_half(tot_r) + 1e-9f);
        float accl[C], accr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { accl[c] = 0.0f; accr[c] = 0.0f; }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const hpair wpair = half_pair(wl[it] * inv_l, wr[it] * inv_r);
            const pvec v = pp[ptap[it]];
#pragma unroll
            for (int c = 0; c < C; ++c) mul_acc_half(v[c], wpair, accl[c], accr[c]);
        }
        float q, s2;
        wave_sum6(accl, accr, q, s2);
        if (store_q) oq[x] = round_half(q);
        if (store_s) os[x] = round_half(s2);
    };
#pragma unroll 1
    for (int j = 0; j < PPW; j += 2) {
        const int x = x0 + wave + j * NW;
        load_w(x + NW, wb, rb);
        pixel(x, wa, ra);
        if (j + 2 < PPW) load_w(x + 2 * NW, wa, ra);
        pixel(x + NW, wb, rb);
    }
}

// ---------------------------------------------------------------------------
// diagnostics: does the Lean math policy ever differ from IEEE?
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

// mode 0: sqrt over EVERY fp32 bit pattern i in [first, first+count)  (exhaustive when
//         first = 0, count = 2^32);
// mode 1: division over pseudo-random operand pairs: mantissas uniform over all 2^23 values,
//         exponents uniform in [-exp_span, exp_span], random signs;
// mode 3: sqrt_pos over every fp32 bit pattern i in [first, first+count) inside [2^-100, 2^100];
// mode 2: division over mantissa pairs i in [first, first+count) of the 2^46 pairs
//         (a = 1.m_a, b = 1.m_b; exhaustive when first = 0, count = 2^46).
// out[0] = number of results whose bits differ from the IEEE result, out[1..] = up to 8
// offending operand bit patterns.
__global__ void k_selftest_math(int mode, uint64_t first, uint64_t count, int exp_span,
                                unsigned long long* __restrict__ out)
{
    unsigned long long bad = 0;
    for (uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < first + count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        if (mode == 0) {
            const float x = __uint_as_float((uint32_t)i);
            const float a = Lean::sqrt(x), b = __builtin_sqrtf(x);
            const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
            if (!same) {
                const unsigned long long k = atomicAdd(&out[0], 1ull);
                if (k < 8) out[1 + k] = i;
            }
        } else if (mode == 3) {
            // sqrt_pos (rsq + Markstein correction) over every fp32 bit pattern in the range it
            // is specified for, [2^-100, 2^100]
            const float x = __uint_as_float((uint32_t)i);
            if (x >= 0x1p-100f && x <= 0x1p100f &&
                __float_as_uint(Lean::sqrt_pos(x)) != __float_as_uint(__builtin_sqrtf(x))) {
                const unsigned long long k = atomicAdd(&out[0], 1ull);
                if (k < 8) out[1 + k] = i;
            }
        } else if (mode == 2) {
            // exhaustive division: i enumerates ALL mantissa pairs, operands in [1, 2)
            const float a = __uint_as_float(0x3f800000u | (uint32_t)(i & 0x7fffffu));
            const float b = __uint_as_float(0x3f800000u | (uint32_t)((i >> 23) & 0x7fffffu));
            if (__float_as_uint(Lean::div(a, b)) != __float_as_uint(a / b)) {
                const unsigned long long k = atomicAdd(&out[0], 1ull);
                if (k < 8) out[1 + k] = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(b);
            }
        } else {
            const uint32_t h0 = mix32(2 * i + 1), h1 = mix32(2 * i + 2), h2 = mix32(~i);
            const int ea = 127 + (int)(h2 % (2 * exp_span + 1)) - exp_span;
            const int eb = 127 + (int)((h2 >> 8) % (2 * exp_span + 1)) - exp_span;
            const float a = __uint_as_float((h0 & 0x807fffffu) | ((uint32_t)ea << 23));
            const float b = __uint_as_float((h1 & 0x807fffffu) | ((uint32_t)eb << 23));
            const float q = Lean::div(a, b);
            if (__float_as_uint(q) != __float_as_uint(a / b)) {
                ++bad;
                const unsigned long long k = atomicAdd(&out[0], 1ull);
                if (k < 8) out[1 + k] = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(b);
            }
        }
    }
    (void)bad;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline int grid_for(int64_t work, int block, int cap = 256 * 16)
{
    int64_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int sdirt_abi_version(void) { return SDIRT_ABI_VERSION; }

const char* sdirt_last_error(void) { return g_err; }

int sdirt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sdirt_lens_create(const sdirt_surface_desc* surfaces, int32_t n_surfaces, sdirt_lens** out)
{
    if (!surfaces || !out) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_surfaces < 1 || n_surfaces > SDIRT_MAX_SURFACES)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "n_surfaces=%d outside [1,%d]", n_surfaces,
                    SDIRT_MAX_SURFACES);
    for (int i = 0; i < n_surfaces; ++i) {
        const sdirt_surface_desc& s = surfaces[i];
        if (s.kind < SDIRT_PLANE || s.kind > SDIRT_ASPHERE)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface %d: unknown kind %d", i, s.kind);
        if (s.ai_degree < 0 || s.ai_degree > SDIRT_MAX_AI)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface %d: ai_degree %d outside [0,%d]", i,
                        s.ai_degree, SDIRT_MAX_AI);
        if ((s.kind == SDIRT_PLANE) != (s.c == 0.0f))
            return fail(SDIRT_ERR_INVALID_ARGUMENT,
                        "surface %d: kind/curvature mismatch (plane <=> c == 0)", i);
        if (!(s.n1 > 0.0) || !(s.n2 > 0.0))
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface %d: refractive index <= 0", i);
    }
    sdirt_lens* L = new (std::nothrow) sdirt_lens();
    if (!L) return fail(SDIRT_ERR_HIP, "out of host memory");
    L->n_surfaces = n_surfaces;
    L->dev = nullptr;
    L->host.resize(n_surfaces);
    for (int i = 0; i < n_surfaces; ++i) L->host[i] = make_dev_surface(surfaces[i]);
    hipError_t e = hipMalloc(&L->dev, sizeof(DevSurface) * n_surfaces);
    if (e == hipSuccess)
        e = hipMemcpy(L->dev, L->host.data(), sizeof(DevSurface) * n_surfaces,
                      hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (L->dev) (void)hipFree(L->dev);
        delete L;
        return fail(e == hipErrorNoDevice ? SDIRT_ERR_NO_DEVICE : SDIRT_ERR_HIP,
                    "lens upload failed: %s", hipGetErrorString(e));
    }
    *out = L;
    return SDIRT_OK;
}

void sdirt_lens_destroy(sdirt_lens* lens)
{
    if (!lens) return;
    if (lens->dev) (void)hipFree(lens->dev);
    delete lens;
}

int32_t sdirt_lens_num_surfaces(const sdirt_lens* lens) { return lens ? lens->n_surfaces : 0; }

// Source of the per-prescription specialisation of the fused kernel: the constants of the two
// device tables (primary wavelength, chief-ray wavelength) as literals, one accessor type per
// surface with the interface of sdirt::Surf, and trace_spec<CHIEF>() = the surface loop unrolled.
// The text is compiled by tools/spec_build.py (-DSDIRT_SPEC_HEADER=...) into a library whose
// fused kernel is valid for THIS prescription only.  Returns the length of the text (written if
// it fits `cap`).
static void emit_surface(std::string& o, const char* tag, int k, const DevSurface& d)
{
    char b[256];
    auto u = [](float f) { uint32_t v; std::memcpy(&v, &f, 4); return v; };
    const SurfHot& h = d.h;
    const int deg = (int)((h.flags >> 8) & 15u);
    snprintf(b, sizeof b, "struct Spec%sP%d {\n", tag, k); o += b;
    for (const char* nm : {"ai", "kai"}) {
        snprintf(b, sizeof b, "    __device__ __forceinline__ float %s(int i) const {\n        switch (i) {\n", nm); o += b;
        for (int i = 0; i < kMaxAi; ++i) {
            snprintf(b, sizeof b, "        case %d: return __uint_as_float(0x%08xu);\n", i,
                     u(nm[0] == 'a' ? d.p.ai[i] : d.p.kai[i]));
            o += b;
        }
        o += "        default: return 0.0f;\n        }\n    }\n";
    }
    o += "};\n";
    snprintf(b, sizeof b, "struct Spec%sS%d {\n    static constexpr uint32_t kF = 0x%xu;\n", tag, k, h.flags); o += b;
    o += "    __device__ __forceinline__ int kind() const { return (int)(kF & 3u); }\n"
         "    __device__ __forceinline__ int ai_degree() const { return (int)((kF >> 8) & 15u); }\n"
         "    __device__ __forceinline__ bool do_refract() const { return (kF & kFlagRefract) != 0u; }\n"
         "    __device__ __forceinline__ bool k_gt_m1() const { return (kF & kFlagKgtM1) != 0u; }\n"
         "    __device__ __forceinline__ bool c_pos() const { return (kF & kFlagCpos) != 0u; }\n"
         "    __device__ __forceinline__ bool unit_k() const { return (kF & kFlagUnitK) != 0u; }\n";
    const struct { const char* n; float v; } f[] = {
        {"d", h.d}, {"c", h.c}, {"c2", h.c2}, {"onepk", h.onepk}, {"lim_loose", h.lim_loose}, {"r2_lim", h.r2_lim},
        {"lim_tight", h.lim_tight}, {"r_lim", h.lim_tight}, {"d_plus_R", h.d_plus_R}, {"eta", h.eta_f}, {"eta2", h.eta2_f}};
    for (const auto& e : f) {
        snprintf(b, sizeof b, "    __device__ __forceinline__ float %s() const { return __uint_as_float(0x%08xu); }\n", e.n, u(e.v));
        o += b;
    }
    snprintf(b, sizeof b, "    __device__ __forceinline__ Spec%sP%d poly(const DevSurface*) const { return Spec%sP%d{}; }\n};\n",
             tag, k, tag, k);
    o += b;
    (void)deg;
}

int64_t sdirt_emit_spec(const sdirt_surface_desc* primary, const sdirt_surface_desc* center, int32_t K, char* out,
                        int64_t cap)
{
    if (!primary || !center || K < 1 || K > SDIRT_MAX_SURFACES) return -1;
    struct { std::vector<DevSurface> host; } l_, c_;
    for (int k = 0; k < K; ++k) { l_.host.push_back(make_dev_surface(primary[k])); c_.host.push_back(make_dev_surface(center[k])); }
    const auto *lens = &l_, *lens_center = &c_;
    std::string o = "// generated by sdirt_lens_emit_spec: do not edit\nnamespace sdirt {\n";
    for (int k = 0; k < K; ++k) emit_surface(o, "A", k, lens->host[k]);
    for (int k = 0; k < K; ++k) emit_surface(o, "C", k, lens_center->host[k]);
    o += "constexpr int kSpecSurfaces = " + std::to_string(K) + ";\n"
         "// the surface loop of trace_ray, unrolled: tw = the launch's trip table (one signed byte per surface)\n"
         "template <bool CHIEF, class M>\n"
         "__device__ __forceinline__ void trace_spec(const u32x16& tw, Ray& r, uint32_t* lds_mask)\n{\n"
         "    uint32_t m;\n";
    char b[512];
    for (int k = 0; k < K; ++k) {
        snprintf(b, sizeof b,
                 "    if (CHIEF) m = surface_reaction<true, M>(SpecCS%d{}, nullptr, (int)(int8_t)(tw[%d] >> %d), r, [] {});\n"
                 "    else m = surface_reaction<true, M>(SpecAS%d{}, nullptr, (int)(int8_t)(tw[%d] >> %d), r, [] {});\n"
                 "    if (lds_mask && m != 0u) ::lds_or_first_lane(&lds_mask[%d], m);\n",
                 k, k >> 2, (k & 3) * 8, k, k >> 2, (k & 3) * 8, k);
        o += b;
    }
    o += "}\n}  // namespace sdirt\n";
    if (out && (int64_t)o.size() + 1 <= cap) std::memcpy(out, o.c_str(), o.size() + 1);
    return (int64_t)o.size() + 1;
}

int sdirt_points_to_object(const float* points, int64_t N, double tan_hfov, double r_last,
                           double sensor_w, double sensor_h, float* point_obj, void* stream)
{
    if (!points || !point_obj || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (N == 0) return SDIRT_OK;
    k_points_to_object<<<grid_for(N, kBlock, 1 << 30), kBlock, 0, as_stream(stream)>>>(
        points, N, (float)tan_hfov, (float)r_last, (float)sensor_w, (float)sensor_h, point_obj);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_pupil_samples(const float* u_theta, const float* u_r2, int64_t S, double pupil_r,
                        float* x2, float* y2, void* stream)
{
    if (!u_theta || !u_r2 || !x2 || !y2 || S < 0)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (S == 0) return SDIRT_OK;
    k_pupil_samples<<<grid_for(S, kBlock, 1 << 30), kBlock, 0, as_stream(stream)>>>(
        u_theta, u_r2, S, (float)(pupil_r * pupil_r), x2, y2);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

static int check_rays(const sdirt_rays& R)
{
    if (!R.ox || !R.oy || !R.oz || !R.dx || !R.dy || !R.dz || !R.ra)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "sdirt_rays has a null array");
    return SDIRT_OK;
}

int sdirt_sample_rays(const float* point_obj, int64_t N, const float* x2, const float* y2, int64_t S,
                      double pupil_z, sdirt_rays rays, void* stream)
{
    if (!point_obj || !x2 || !y2 || N < 0 || S < 0)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_rays(rays)) return rc;
    if (N * S == 0) return SDIRT_OK;
    k_sample_rays<<<grid_for(N * S, kBlock), kBlock, 0, as_stream(stream)>>>(
        point_obj, N, x2, y2, S, (float)pupil_z, rays);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_rays_from_aos(const float* o, const float* d, const float* ra, int64_t M, int32_t normalize,
                        sdirt_rays rays, void* stream)
{
    if (!o || !d || M < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_rays(rays)) return rc;
    if (M == 0) return SDIRT_OK;
    k_rays_from_aos<<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>(o, d, ra, M, normalize,
                                                                           rays);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_rays_to_aos(sdirt_rays rays, int64_t M, float* o, float* d, void* stream)
{
    if (M < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "n_rays < 0");
    if (int rc = check_rays(rays)) return rc;
    if (M == 0 || (!o && !d)) return SDIRT_OK;
    k_rays_to_aos<<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>(rays, M, o, d);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_trace(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                const int32_t* trips, uint32_t flags, sdirt_rays rays, int64_t M,
                uint32_t* conv_mask, void* stream)
{
    if (!lens) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null lens");
    if (first < 0 || last > lens->n_surfaces || first > last)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface range [%d,%d) outside [0,%d]", first, last,
                    lens->n_surfaces);
    if (int rc = check_rays(rays)) return rc;
    TripTable tt;
    if (int rc = make_trips(lens, trips, tt)) return rc;
    if (M <= 0 || first == last) return M < 0 ? fail(SDIRT_ERR_INVALID_ARGUMENT, "n_rays < 0") : SDIRT_OK;
    const int grid = grid_for(M, kBlock);
    const bool lean = (flags & SDIRT_PSF_STRICT_IEEE) == 0;
#define SDIRT_LAUNCH_TRACE(FW, MM)                                                              \
    k_trace<FW, MM><<<grid, kBlock, 0, as_stream(stream)>>>(tt, lens->dev, lens->n_surfaces,    \
                                                            first, last, rays, M, conv_mask)
    if (backward) {
        if (lean) SDIRT_LAUNCH_TRACE(false, Lean); else SDIRT_LAUNCH_TRACE(false, Ieee);
    } else {
        if (lean) SDIRT_LAUNCH_TRACE(true, Lean); else SDIRT_LAUNCH_TRACE(true, Ieee);
    }
#undef SDIRT_LAUNCH_TRACE
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_propagate_to(double z, sdirt_rays rays, int64_t M, void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (M <= 0) return M < 0 ? fail(SDIRT_ERR_INVALID_ARGUMENT, "n_rays < 0") : SDIRT_OK;
    k_propagate<<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>((float)z, rays, M);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_center_from_rays(sdirt_rays rays, int64_t S, int64_t N, float* center, int32_t* any_valid,
                           void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (!center || S < 0 || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (N == 0) return SDIRT_OK;
    k_center_from_rays<<<grid_for(N, 64, 1 << 30), 64, 0, as_stream(stream)>>>(rays, S, N, center,
                                                                               any_valid);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

static int check_ks(int ks)
{
    if (ks < 2 || ks > SDIRT_MAX_KS)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "ks=%d outside [2,%d]", ks, SDIRT_MAX_KS);
    return SDIRT_OK;
}

int sdirt_forward_integral(sdirt_rays rays, int64_t S, int64_t N, double ps, int32_t ks,
                           const float* center, const sdirt_dp_params* dp, float* l_grid,
                           float* r_grid, void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (int rc = check_ks(ks)) return rc;
    if (!center || !l_grid || S < 0 || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    if (N == 0) return SDIRT_OK;
    const size_t bytes = sizeof(float) * (size_t)N * ks * ks;
    HIP_TRY(hipMemsetAsync(l_grid, 0, bytes, as_stream(stream)));
    if (r_grid) HIP_TRY(hipMemsetAsync(r_grid, 0, bytes, as_stream(stream)));
    if (S == 0) return SDIRT_OK;
    k_forward_integral<<<grid_for(S * N, kBlock), kBlock, 0, as_stream(stream)>>>(
        rays, S, N, make_geom(ps, ks), make_dp(dp), center, l_grid, r_grid);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_psf_normalize(float* psf, int64_t N, int32_t ks, void* stream)
{
    if (!psf || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (ks < 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "ks < 1");
    if (N == 0) return SDIRT_OK;
    k_psf_normalize<<<(int)N, kBlock, 0, as_stream(stream)>>>(psf, ks * ks);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_chief_center(const sdirt_lens* lens, const float* point_obj, int64_t N, const float* xc,
                       const float* yc, int64_t Sc, double pupil_z, double d_sensor,
                       const int32_t* trips, uint32_t flags, float* center, int32_t* any_valid,
                       uint32_t* conv_mask, void* stream)
{
    if (!lens || !point_obj || !xc || !yc || !center || N < 0 || Sc < 0 || Sc > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    TripTable tt;
    if (int rc = make_trips(lens, trips, tt)) return rc;
    if (N == 0) return SDIRT_OK;
    if (!(flags & SDIRT_PSF_STRICT_IEEE))
        k_chief_center<Lean><<<(int)N, kFused, 0, as_stream(stream)>>>(
            tt, lens->dev, lens->n_surfaces, point_obj, xc, yc, (int)Sc, (float)pupil_z,
            (float)d_sensor, center, any_valid, conv_mask);
    else
        k_chief_center<Ieee><<<(int)N, kFused, 0, as_stream(stream)>>>(
            tt, lens->dev, lens->n_surfaces, point_obj, xc, yc, (int)Sc, (float)pupil_z,
            (float)d_sensor, center, any_valid, conv_mask);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

// Shared launcher of sdirt_psf_lr / sdirt_psf_lr_centered / sdirt_psf_rgb_centered.  `cen` != nullptr
// requests the chief-ray pass: inside the same kernel when one workgroup owns a point (nsplit ==
// 1), as a preceding k_chief_center launch otherwise.  W wavelength slots (W > 1 needs nsplit == 1):
// lens[w], trips[w], x2 / y2 [W][S], outputs [N][W][ks][ks], masks [W][SDIRT_MAX_SURFACES].
struct CenterRequest {
    const sdirt_lens* lens_c;
    const float* xc;              // [W][Sc]
    const float* yc;
    int64_t Sc;
    TripSet trips_c;
    float* center_out;            // [W][N][2]
    int32_t* any_valid;           // [W]
    uint32_t* conv_mask_c;        // [W][SDIRT_MAX_SURFACES]
};

static int spp_split(int64_t N, int64_t S, int* chunk_out)
{
    // Fill the chip: at least ~4 workgroups per CU; split the spp axis when the
    // number of points alone cannot (e.g. PSFNet training: N=64, S=20000).
    int nsplit = 1;
    const int64_t want_blocks = 256 * 4;
    if (N < want_blocks && S > 2 * kFused) {
        nsplit = (int)((want_blocks + N - 1) / N);
        const int max_split = (int)((S + 2 * kFused - 1) / (2 * kFused));
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    }
    int chunk = (int)((S + nsplit - 1) / nsplit);
    chunk = ((chunk + kFused - 1) / kFused) * kFused;
    nsplit = (int)((S + chunk - 1) / (chunk > 0 ? chunk : 1));
    if (nsplit < 1) nsplit = 1;
    if (chunk_out) *chunk_out = chunk;
    return nsplit;
}

static int launch_psf(const sdirt_lens* const* lens, int W, const float* point_obj, int64_t N,
                      const float* x2, const float* y2, int64_t S, double pupil_z, double d_sensor,
                      double ps, int32_t ks, const float* center, const CenterRequest* cen,
                      const sdirt_dp_params* dp, const TripSet& tt, uint32_t flags, float* l_psf,
                      float* r_psf, uint32_t* conv_mask, void* stream)
{
    const bool have_r = r_psf != nullptr;
    const int tile = ks * ks;
    const size_t lds = sizeof(float) * tile * (have_r ? 2 : 1);
    // a multi-wavelength launch keeps one workgroup per (point, wavelength): the chief-ray pass
    // stays fused and the whole of psf_rgb is one kernel, also for the few points of a psf_map
    int chunk = ((int)S + kFused - 1) / kFused * kFused;
    const int nsplit = W > 1 ? 1 : spp_split(N, S, &chunk);
    const int K = lens[0]->n_surfaces;

    hipStream_t st = as_stream(stream);
    const bool lean = (flags & SDIRT_PSF_STRICT_IEEE) == 0;
    const bool fuse_center = cen != nullptr && nsplit == 1;
    if (cen && !fuse_center) {                       // split spp axis: centre as its own launch
        if (lean)
            k_chief_center<Lean><<<(int)N, kFused, 0, st>>>(
                cen->trips_c.t[0], cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc,
                (float)pupil_z, (float)d_sensor, cen->center_out, cen->any_valid, cen->conv_mask_c);
        else
            k_chief_center<Ieee><<<(int)N, kFused, 0, st>>>(
                cen->trips_c.t[0], cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc,
                (float)pupil_z, (float)d_sensor, cen->center_out, cen->any_valid, cen->conv_mask_c);
        LAUNCH_CHECK();
        center = cen->center_out;
    }
    if (nsplit > 1) {
        HIP_TRY(hipMemsetAsync(l_psf, 0, sizeof(float) * (size_t)N * tile, st));
        if (have_r) HIP_TRY(hipMemsetAsync(r_psf, 0, sizeof(float) * (size_t)N * tile, st));
    }
    const SplatGeom gm = make_geom(ps, ks);
    const DevDpParams dpp = make_dp(dp);
    const SplatBlock sblk = make_splat_block(gm, dpp);
    const dim3 grid((unsigned)(N * nsplit), (unsigned)W);
    const bool both = have_r && dpp.have_r;
    size_t lds_bytes = both ? lds : sizeof(float) * tile;
    CenterArgs ca;
    TripSet ttc;
    LensSet ls;
    std::memset(&ca, 0, sizeof(ca));
    std::memset(&ttc, 0, sizeof(ttc));
    std::memset(&ls, 0, sizeof(ls));
    for (int w = 0; w < W; ++w) ls.p[w] = lens[w]->dev;
    if (fuse_center) {
        ca.lens_c = cen->lens_c->dev; ttc = cen->trips_c; ca.xc = cen->xc; ca.yc = cen->yc;
        ca.Sc = (int)cen->Sc; ca.center_out = cen->center_out; ca.any_valid = cen->any_valid;
        ca.conv_mask_c = cen->conv_mask_c;
        lds_bytes = std::max(lds_bytes, sizeof(double) * 3 * kFused);   // fp64 reduction scratch
    }
#define SDIRT_LAUNCH_PSF(HR, BG, MM, CT)                                                          \
    do {                                                                                          \
        if (lds_bytes > 48 * 1024) /* large tiles: opt in to the full 160 KiB of LDS */           \
            HIP_TRY(hipFuncSetAttribute((const void*)k_psf_lr<HR, BG, MM, CT>,                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                        160 * 1024 - 1024));                                      \
        k_psf_lr<HR, BG, MM, CT><<<grid, kFused, lds_bytes, st>>>(                                \
            sblk, tt, ttc, ls, K, point_obj, x2, y2, (int)S, nsplit, chunk, (float)pupil_z,       \
            (float)d_sensor, ks, dpp.tr, dpp.tl, center, flags, l_psf, both ? r_psf : nullptr,    \
            conv_mask, ca);                                                                       \
    } while (0)
#define SDIRT_LAUNCH_PSF_C(HR, BG, MM)                                                            \
    do {                                                                                          \
        if (fuse_center) SDIRT_LAUNCH_PSF(HR, BG, MM, true); else SDIRT_LAUNCH_PSF(HR, BG, MM, false); \
    } while (0)
#define SDIRT_LAUNCH_PSF_M(HR, BG)                                                                \
    do {                                                                                          \
        if (lean) SDIRT_LAUNCH_PSF_C(HR, BG, Lean); else SDIRT_LAUNCH_PSF_C(HR, BG, Ieee);        \
    } while (0)
    if (both) {
        if (dpp.big) SDIRT_LAUNCH_PSF_M(true, true); else SDIRT_LAUNCH_PSF_M(true, false);
    } else {
        if (dpp.big) SDIRT_LAUNCH_PSF_M(false, true); else SDIRT_LAUNCH_PSF_M(false, false);
        // param_list=None leaves the R grid all-zero (monte_carlo.py:230-235)
        if (have_r) HIP_TRY(hipMemsetAsync(r_psf, 0, sizeof(float) * (size_t)N * W * tile, st));
    }
#undef SDIRT_LAUNCH_PSF_M
#undef SDIRT_LAUNCH_PSF_C
#undef SDIRT_LAUNCH_PSF
    LAUNCH_CHECK();
    if (nsplit > 1 && (flags & SDIRT_PSF_NORMALIZE)) {
        k_psf_normalize<<<(int)N, kBlock, 0, st>>>(l_psf, tile);
        if (have_r && dpp.have_r) k_psf_normalize<<<(int)N, kBlock, 0, st>>>(r_psf, tile);
        LAUNCH_CHECK();
    }
    return SDIRT_OK;
}

int sdirt_psf_lr(const sdirt_lens* lens, const float* point_obj, int64_t N, const float* x2,
                 const float* y2, int64_t S, double pupil_z, double d_sensor, double ps, int32_t ks,
                 const float* center, const sdirt_dp_params* dp, const int32_t* trips,
                 uint32_t flags, float* l_psf, float* r_psf, uint32_t* conv_mask, void* stream)
{
    if (!lens || !point_obj || !x2 || !y2 || !center || !l_psf || N < 0 || S < 0 ||
        S > (1ll << 30) || N > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_ks(ks)) return rc;
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    TripSet tt;
    std::memset(&tt, 0, sizeof(tt));
    if (int rc = make_trips(lens, trips, tt.t[0])) return rc;
    if (N == 0) return SDIRT_OK;
    return launch_psf(&lens, 1, point_obj, N, x2, y2, S, pupil_z, d_sensor, ps, ks, center, nullptr, dp,
                      tt, flags, l_psf, r_psf, conv_mask, stream);
}

int sdirt_psf_lr_centered(const sdirt_lens* lens, const sdirt_lens* lens_center,
                          const float* point_obj, int64_t N, const float* x2, const float* y2,
                          int64_t S, const float* xc, const float* yc, int64_t Sc, double pupil_z,
                          double d_sensor, double ps, int32_t ks, const sdirt_dp_params* dp,
                          const int32_t* trips, const int32_t* trips_center, uint32_t flags,
                          float* center, int32_t* any_valid, float* l_psf, float* r_psf,
                          uint32_t* conv_mask, uint32_t* conv_mask_center, void* stream)
{
    return sdirt_psf_rgb_centered(&lens, 1, lens_center, point_obj, N, x2, y2, S, xc, yc, Sc, pupil_z,
                                  d_sensor, ps, ks, dp, trips, trips_center, flags, center, any_valid,
                                  l_psf, r_psf, conv_mask, conv_mask_center, stream);
}

int sdirt_psf_rgb_centered(const sdirt_lens* const* lens, int32_t W, const sdirt_lens* lens_center,
                           const float* point_obj, int64_t N, const float* x2, const float* y2,
                           int64_t S, const float* xc, const float* yc, int64_t Sc, double pupil_z,
                           double d_sensor, double ps, int32_t ks, const sdirt_dp_params* dp,
                           const int32_t* trips, const int32_t* trips_center, uint32_t flags,
                           float* center, int32_t* any_valid, float* l_psf, float* r_psf,
                           uint32_t* conv_mask, uint32_t* conv_mask_center, void* stream)
{
    if (!lens || W < 1 || W > SDIRT_MAX_WAVELENGTHS || !lens_center || !point_obj || !x2 || !y2 || !xc ||
        !yc || !center || !l_psf || N < 0 || S < 0 || Sc < 0 || S > (1ll << 30) || Sc > (1ll << 30) ||
        N > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    for (int w = 0; w < W; ++w)
        if (!lens[w] || lens[w]->n_surfaces != lens_center->n_surfaces)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "lens[%d] missing or surface count differs from lens_center", w);
    if (int rc = check_ks(ks)) return rc;
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    const int K = lens_center->n_surfaces;
    TripSet tt;
    CenterRequest cr;
    std::memset(&tt, 0, sizeof(tt));
    std::memset(&cr.trips_c, 0, sizeof(cr.trips_c));
    for (int w = 0; w < W; ++w) {
        if (int rc = make_trips(lens[w], trips ? trips + (size_t)w * K : nullptr, tt.t[w])) return rc;
        if (int rc = make_trips(lens_center, trips_center ? trips_center + (size_t)w * K : nullptr,
                                cr.trips_c.t[w]))
            return rc;
    }
    if (N == 0) return SDIRT_OK;
    cr.lens_c = lens_center; cr.xc = xc; cr.yc = yc; cr.Sc = Sc; cr.center_out = center;
    cr.any_valid = any_valid; cr.conv_mask_c = conv_mask_center;
    return launch_psf(lens, W, point_obj, N, x2, y2, S, pupil_z, d_sensor, ps, ks, nullptr, &cr, dp, tt,
                      flags, l_psf, r_psf, conv_mask, stream);
}

int sdirt_selftest_math(int32_t mode, uint64_t first, uint64_t count, int32_t exp_span,
                        uint64_t* out, void* stream)
{
    if (!out || mode < 0 || mode > 3 || exp_span < 0 || exp_span > 60)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(uint64_t) * 9, as_stream(stream)));
    if (count == 0) return SDIRT_OK;
    k_selftest_math<<<256 * 32, 256, 0, as_stream(stream)>>>(mode, first, count, exp_span,
                                                            (unsigned long long*)out);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_local_psf_render(const float* img, const float* psf, int32_t B, int32_t C, int32_t H,
                           int32_t W, int32_t ks, int32_t half_precision, float* out_l, float* out_r,
                           void* stream)
{
    if (!img || !psf || !out_l || !out_r || B < 0 || H < 1 || W < 1 || ks < 1 || (ks & 1) == 0)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument (ks must be odd)");
    if (B == 0) return SDIRT_OK;
    const int64_t P = (int64_t)B * H * W;
    const int grid = grid_for(P * 64, kBlock, 256 * 32);      // one wave per pixel, grid-stride
    hipStream_t st = as_stream(stream);
    // row-mapped LDS-tiled kernel whenever 8 (or 4, or 2) pixels' kernels fit in 64 KB of LDS and the
    // image fits 32-bit offsets, else the direct one-wave-per-pixel kernel
    const size_t per_pixel = sizeof(float) * 2 * (size_t)ks * ks;
    const bool small = ks <= 64 && (int64_t)C * H * W < (1ll << 30) && (int64_t)B * H < 65536;
    const int pix = !small ? 0 : per_pixel * 8 + 16 <= 64 * 1024 ? 8 : per_pixel * 4 + 16 <= 64 * 1024 ? 4
                    : per_pixel * 2 + 16 <= 64 * 1024 ? 2 : 0;
    const size_t lds_tile = per_pixel * pix + 16;
    const int groups = pix ? (W + pix - 1) / pix : 0;
    // ~16 workgroups per CU in flight; the rest of a row's groups are walked by the same workgroup
    const int gx = pix ? std::max(1, std::min(groups, (int)((256 * 16 + (int64_t)B * H - 1) / ((int64_t)B * H)))) : 0;
    const dim3 grid_t((unsigned)gx, (unsigned)(B * H));
    // pipelined kernel (ks 21, RGB): each workgroup walks >= ~6 groups of its row so that the prologue
    // fetch is amortised; LDS = shifted kernel run + image patch
#ifndef SDIRT_PIPE_PIX
#define SDIRT_PIPE_PIX 8
#endif
    const int groups_p = (W + SDIRT_PIPE_PIX - 1) / SDIRT_PIPE_PIX;
    const dim3 grid_p((unsigned)std::max(1, std::min(groups_p, std::max(gx, (groups_p + 11) / 12))), (unsigned)(B * H));
    const size_t lds_pipe = sizeof(float) * (((SDIRT_PIPE_PIX * 2 * 441 + 4 + 3) & ~3) + 3 * 21 * (SDIRT_PIPE_PIX + 20));
    const int64_t total_floats = P * 2 * ks * ks;
    // wave-per-pixel kernel: one workgroup per 64-pixel (SDIRT_RENDER_CHUNK=128: 128-pixel) stretch of a row
    static const int chunk_w = getenv("SDIRT_RENDER_CHUNK") ? atoi(getenv("SDIRT_RENDER_CHUNK")) : 64;
    static const bool use_pipe = getenv("SDIRT_RENDER_PIPE") != nullptr;
    const dim3 grid_w((unsigned)((W + chunk_w - 1) / chunk_w), (unsigned)(B * H));
    auto lds_wave = [&](bool hf) { return (size_t)21 * (chunk_w + 20) * 4 * (hf ? 2 : 4); };
#define SDIRT_RENDER_T(CC, HF, PP)                                                               \
    do {                                                                                         \
        if (lds_tile > 48 * 1024)                                                                \
            HIP_TRY(hipFuncSetAttribute((const void*)k_local_psf_render_rows<CC, HF, PP, 0>,     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
        k_local_psf_render_rows<CC, HF, PP, 0><<<grid_t, kBlock, lds_tile, st>>>(                \
            img, psf, H, W, ks, out_l, out_r);                                                   \
    } while (0)
#define SDIRT_RENDER_H(CC, HF)                                                                   \
    do {                                                                                         \
        /* the reference's PSFNet kernel size (configs/dfdp_by_sdirt_rf50mm.yml: ks 21) on RGB */ \
        if (pix == 8 && ks == 21 && CC == 3 && !use_pipe && chunk_w == 128)                      \
            k_local_psf_render_wave<3, HF, 21, 128><<<grid_w, kBlock, lds_wave(HF), st>>>(       \
                img, psf, H, W, out_l, out_r);                                                   \
        else if (pix == 8 && ks == 21 && CC == 3 && !use_pipe)                                   \
            k_local_psf_render_wave<3, HF, 21, 64><<<grid_w, kBlock, lds_wave(HF), st>>>(        \
                img, psf, H, W, out_l, out_r);                                                   \
        else if (pix == 8 && ks == 21 && CC == 3)                                                \
            k_local_psf_render_pipe<3, HF, 21, SDIRT_PIPE_PIX><<<grid_p, kBlock, lds_pipe, st>>>(\
                img, psf, H, W, total_floats, out_l, out_r);                                     \
        else if (pix == 8) SDIRT_RENDER_T(CC, HF, 8);                                            \
        else if (pix == 4) SDIRT_RENDER_T(CC, HF, 4);                                            \
        else if (pix == 2) SDIRT_RENDER_T(CC, HF, 2);                                            \
        else k_local_psf_render<CC, HF><<<grid, kBlock, 0, st>>>(img, psf, B, H, W, ks, out_l,    \
                                                                out_r);                          \
    } while (0)
#define SDIRT_RENDER(CC)                                                                         \
    do {                                                                                         \
        if (half_precision) SDIRT_RENDER_H(CC, true); else SDIRT_RENDER_H(CC, false);            \
    } while (0)
    switch (C) {
    case 1: SDIRT_RENDER(1); break;
    case 3: SDIRT_RENDER(3); break;
    case 4: SDIRT_RENDER(4); break;
    default: return fail(SDIRT_ERR_UNSUPPORTED, "channels=%d (supported: 1, 3, 4)", C);
    }
#undef SDIRT_RENDER
#undef SDIRT_RENDER_H
#undef SDIRT_RENDER_T
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_psfnet_render(const float* img, const void* raw_l, const void* raw_r, int32_t B, int32_t C,
                        int32_t H, int32_t W, int32_t ks, float* out_l, float* out_r, void* stream)
{
    if (!img || !raw_l || !raw_r || !out_l || !out_r || B < 0 || H < 1 || W < 1 || ks < 1 ||
        (ks & 1) == 0)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument (ks must be odd)");
    if (((uintptr_t)raw_l | (uintptr_t)raw_r) & 15)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "raw_l / raw_r must be 16-byte aligned");
    if (B == 0) return SDIRT_OK;
    const int64_t P = (int64_t)B * H * W;
    hipStream_t st = as_stream(stream);
    const size_t per_pixel = sizeof(_Float16) * 2 * (size_t)ks * ks;
    if (per_pixel * 8 > 64 * 1024)
        return fail(SDIRT_ERR_UNSUPPORTED, "ks=%d: eight pixels' kernels exceed 64 KB of LDS", ks);
    const _Float16* rl = static_cast<const _Float16*>(raw_l);
    const _Float16* rr = static_cast<const _Float16*>(raw_r);
    static const bool use_pipe = getenv("SDIRT_RENDER_PIPE") != nullptr;
#define SDIRT_PN(CC, PP, KK)                                                                     \
    do {                                                                                         \
        const size_t lds = per_pixel * PP;                                                       \
        const int grid = (int)std::min<int64_t>((P + PP - 1) / PP, 256 * 64);                    \
        if (lds > 48 * 1024)                                                                     \
            HIP_TRY(hipFuncSetAttribute((const void*)k_psfnet_render<CC, PP, KK>,                \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
        k_psfnet_render<CC, PP, KK><<<grid, kBlock, lds, st>>>(img, rl, rr, B, H, W, ks, out_l,  \
                                                               out_r);                           \
    } while (0)
#define SDIRT_PN_C(CC)                                                                           \
    do {                                                                                         \
        if (ks == 21 && CC == 3 && (int64_t)3 * H * W < (1ll << 30) && (int64_t)B * H < 65536 && !use_pipe) { \
            k_psfnet_render_wave<3, 21, 64><<<dim3((unsigned)((W + 63) / 64), (unsigned)(B * H)), kBlock, \
                                              (size_t)21 * 84 * 8, st>>>(img, rl, rr, H, W, out_l, out_r); \
        } else if (ks == 21 && CC == 3 && (int64_t)3 * H * W < (1ll << 30) && (int64_t)B * H < 65536) { \
            const int groups = (W + 7) / 8;                                                      \
            const int gx = std::max(1, std::min(groups, std::max((int)((256 * 16 + (int64_t)B * H - 1) / ((int64_t)B * H)), (groups + 11) / 12))); \
            const size_t lds = sizeof(_Float16) * (2 * ((8 * 441 + 8 + 7) & ~7) + 3 * 21 * 28);  \
            k_psfnet_render_pipe<3, 21><<<dim3((unsigned)gx, (unsigned)(B * H)), kBlock, lds, st>>>( \
                img, rl, rr, H, W, P * 441, out_l, out_r);                                       \
        } else if (ks == 21 && CC == 3) SDIRT_PN(3, 16, 21);                                     \
        else if (per_pixel * 16 <= 32 * 1024) SDIRT_PN(CC, 16, 0);                               \
        else SDIRT_PN(CC, 8, 0);                                                                 \
    } while (0)
    switch (C) {
    case 1: SDIRT_PN_C(1); break;
    case 3: SDIRT_PN_C(3); break;
    case 4: SDIRT_PN_C(4); break;
    default: return fail(SDIRT_ERR_UNSUPPORTED, "channels=%d (supported: 1, 3, 4)", C);
    }
#undef SDIRT_PN_C
#undef SDIRT_PN
    LAUNCH_CHECK();
    return SDIRT_OK;
}

}  // extern "C"
