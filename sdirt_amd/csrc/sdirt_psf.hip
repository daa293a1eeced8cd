// sdirt_psf.hip -- the PSF kernels of libsdirt_dp.so (MI355X / gfx950 only): forward_integral on
// staged rays, max-normalisation, the fused chief-ray centre, and k_psf_lr = [chief-ray pass ->]
// sample -> trace -> propagate -> window -> dual-pixel weights -> LDS splat -> normalise -> store
// (deeplens/optics.py:889-996, deeplens/monte_carlo.py:9-372).  See include/sdirt_dp.h for the
// ABI and DESIGN.md for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/sdirt_dp.h"
#include "sdirt_trace.hpp"

using namespace sdirt;

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

// forward_integral on a point-major SoA bundle, grids too large for LDS (ks > SDIRT_MAX_KS; the one caller is a
// plot that traces three points): one thread per ray, contributions added to the pre-zeroed [N,ks,ks] grids in HBM
// with global float atomics (the 64 adds of a wave instruction fall into ONE tile).
__global__ void __launch_bounds__(kBlock)
k_forward_integral_hbm(sdirt_rays R, int64_t S, int64_t N, SplatGeom gm, DevDpParams dp,
                       const float* __restrict__ center, float* __restrict__ lg,
                       float* __restrict__ rg)
{
    const int64_t M = S * N;
    const int64_t tile = (int64_t)gm.ks * gm.ks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / S;
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

// forward_integral on a point-major SoA bundle with the grids in LDS (monte_carlo.py:9-68 for every ks <=
// SDIRT_MAX_KS).
//
// A workgroup OWNS P consecutive points (one, unless a point has fewer samples than the workgroup has threads)
// and a slice of the spp axis: their L/R tiles live in LDS for the whole kernel (no global atomics, no memset,
// each tile stored once, coalesced) -- or, when the few points of a call are cut along spp (nsplit > 1), added
// once per workgroup to the zeroed output.  Thread t works on sample t % rp of point t / rp (rp = 512 / P): a wave
// reads 64 consecutive samples of one point, 256 contiguous bytes per component (sample-major bundles, the
// reference's tensor order, gave a workgroup 8 to 32 useful bytes of every 128-byte line: 44.8 M L2 requests and
// 352 us for the 16.8 M rays of the staged bench at ks 65, profiles/r04/staged_pmc_sample_major.json).
//
// The tiles are DOUBLES whenever two of them fit (ks <= 99): on gfx950 one ds_add_f32 wave instruction occupies the
// LDS for ~193 cycles whatever its addresses (it is executed lane by lane), ds_add_f64 for 17-30 and ds_add_u64 for
// 8-20 (tools/lds_atomic_bench.hip, profiles/r04/lds_atomic_bench*.txt) -- with eight adds per ray the fp32 form
// IS the kernel's time (350 of 350 us on 8.4 M rays).  A double sum rounded once on the way out is also nearer to
// the reference's sequential fp32 sum's exact value than any fp32 summation order, and takes any weight `ra`.
constexpr int kFiThreads = 512;          // 256 / 512 / 1024 measured on nine batch shapes: 512 is 3-12 % ahead (profiles/r04)
constexpr int kFiDepth = 2;             // passes whose rays are in flight (2, 4, 8 time within 7 %: profiles/r04)
struct FiLaunch {
    int P, logRp;         // points per workgroup; log2(samples per pass and point = kFiThreads / P)
    int ngroups;          // ceil(N / P)
    int nsplit;           // slices of the spp axis
    int64_t chunk;        // samples per slice, a multiple of kFiThreads / P
    int stride;           // accumulators per point in LDS: (ntile * ks * ks) | 1, odd -> the same pixel of different
                          // points never shares a bank
};
template <bool HAVE_R, bool BIG, class ACC, class M>
__global__ void __launch_bounds__(kFiThreads)
k_forward_integral_tiles(sdirt_rays R, int64_t S, int64_t N, SplatGeom gm, DevDpParams dp, FiLaunch fl,
                         const float* __restrict__ center, float* __restrict__ lg, float* __restrict__ rg, int normalize)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fi_lds[];
    __shared__ float fi_red[kFiThreads / 64];
    ACC* __restrict__ fi_tiles = reinterpret_cast<ACC*>(fi_lds);
    const int tile = gm.ks * gm.ks;
    const int g = (int)(blockIdx.x / (uint32_t)fl.nsplit), j = (int)(blockIdx.x - (uint32_t)g * fl.nsplit);
    const int rp = 1 << fl.logRp;
    const int p = threadIdx.x >> fl.logRp, row = threadIdx.x & (rp - 1);
    const int64_t n = (int64_t)g * fl.P + p;
    const bool have_pt = n < N;
    for (int i = threadIdx.x; i < fl.P * fl.stride; i += kFiThreads) fi_tiles[i] = (ACC)0;
    __syncthreads();

    const float cx = have_pt ? center[2 * n] : 0.0f, cy = have_pt ? center[2 * n + 1] : 0.0f;
    ACC* __restrict__ tl_ = fi_tiles + p * fl.stride;
    ACC* __restrict__ trr = tl_ + tile;
    const auto div_dy = UDiv<M>::make(gm.dy_rng), div_dx = UDiv<M>::make(gm.dx_rng);
    const auto div_fmh = UDiv<M>::make(dp.fmh);
    const int64_t s_begin = (int64_t)j * fl.chunk, s_end = min(S, s_begin + fl.chunk);
    const int npass = (int)((s_end - s_begin + rp - 1) >> fl.logRp);     // the same for every thread: no vote, no barrier
    int64_t s = s_begin + row;
    int64_t i = (have_pt ? n : 0) * S + s;
    // The rays of the next kFiDepth passes are in flight while one pass is splatted.  Every thread loads on every
    // pass -- lanes without a ray read element 0 and ignore it -- so that the number of loads in flight is the same on
    // every path: the compiler then waits with a counted s_waitcnt vmcnt for exactly the set it is about to use (a
    // predicated prefetch forces vmcnt(0) right behind its own issue).
    float vox[kFiDepth], voy[kFiDepth], vdx[kFiDepth], vdz[kFiDepth], vra[kFiDepth];
    bool vcur[kFiDepth];
#pragma unroll
    for (int k = 0; k < kFiDepth; ++k) {
        vcur[k] = have_pt && s < s_end;
        const int64_t il = vcur[k] ? i : 0;
        vox[k] = R.ox[il]; voy[k] = R.oy[il]; vdx[k] = R.dx[il]; vdz[k] = R.dz[il]; vra[k] = R.ra[il];
        s += rp; i += rp;
    }
    for (int pass = 0; pass < npass; pass += kFiDepth) {
#pragma unroll
        for (int k = 0; k < kFiDepth; ++k) {
            const float ox = vox[k], oy = voy[k], dx = vdx[k], dz = vdz[k], ra = vra[k];
            const bool cur = vcur[k];
            vcur[k] = have_pt && s < s_end;
            const int64_t il = vcur[k] ? i : 0;
            vox[k] = R.ox[il]; voy[k] = R.oy[il]; vdx[k] = R.dx[il]; vdz[k] = R.dz[il]; vra[k] = R.ra[il];
            s += rp; i += rp;
            SplatTaps tp;
            if (cur && splat_taps(gm, div_dy, div_dx, ox, oy, cx, cy, ra, tp)) {
                const float x_tan = (-dx) / dz;              // monte_carlo.py:48
                float sl, sr;
                if (BIG) dp_weights_big(dp, x_tan, sl, sr);
                else dp_weights_small<M>(dp, div_fmh, x_tan, sl, sr);
                atomicAdd(&tl_[tp.i_tl], (ACC)(tp.w_tl * sl));
                atomicAdd(&tl_[tp.i_tr], (ACC)(tp.w_tr * sl));
                atomicAdd(&tl_[tp.i_bl], (ACC)(tp.w_bl * sl));
                atomicAdd(&tl_[tp.i_br], (ACC)(tp.w_br * sl));
                if (HAVE_R) {
                    atomicAdd(&trr[tp.i_tl], (ACC)(tp.w_tl * sr));
                    atomicAdd(&trr[tp.i_tr], (ACC)(tp.w_tr * sr));
                    atomicAdd(&trr[tp.i_bl], (ACC)(tp.w_bl * sr));
                    atomicAdd(&trr[tp.i_br], (ACC)(tp.w_br * sr));
                }
            }
        }
    }
    __syncthreads();
    // the P tiles of this workgroup are P * ks * ks consecutive floats of the [N, ks, ks] output
    const int np = (int)min((int64_t)fl.P, N - (int64_t)g * fl.P);
    for (int pp = 0; pp < np; ++pp) {
        const ACC* __restrict__ src = fi_tiles + pp * fl.stride;
        float* __restrict__ Lg = lg + ((int64_t)g * fl.P + pp) * tile;
        float* __restrict__ Rg = HAVE_R ? rg + ((int64_t)g * fl.P + pp) * tile : nullptr;
        if (fl.nsplit == 1 && normalize) {
            // optics.py:983-987 on the way out (SDIRT_PSF_NORMALIZE): psf / (max + 1e-6) of the float grids, the same
            // values and the same division k_psf_normalize would read back and perform
            float ml = -INFINITY, mr = -INFINITY;
            for (int e = threadIdx.x; e < tile; e += kFiThreads) {
                ml = fmaxf(ml, (float)src[e]);
                if (HAVE_R) mr = fmaxf(mr, (float)src[tile + e]);
            }
            const float dl = block_max(ml, fi_red) + 1e-6f;
            const float dr = HAVE_R ? block_max(mr, fi_red) + 1e-6f : 1.0f;
            for (int e = threadIdx.x; e < tile; e += kFiThreads) {
                Lg[e] = (float)src[e] / dl;
                if (HAVE_R) Rg[e] = (float)src[tile + e] / dr;
            }
        } else if (fl.nsplit == 1) {
            for (int e = threadIdx.x; e < tile; e += kFiThreads) {
                Lg[e] = (float)src[e];
                if (HAVE_R) Rg[e] = (float)src[tile + e];
            }
        } else {
            for (int e = threadIdx.x; e < tile; e += kFiThreads) {
                const float a = (float)src[e];
                if (a != 0.0f) atomicAdd(&Lg[e], a);
                if (HAVE_R) {
                    const float c = (float)src[tile + e];
                    if (c != 0.0f) atomicAdd(&Rg[e], c);
                }
            }
        }
    }
}

// optics.py:983-987: one workgroup per point.  LDS_TILE: the tile is read from HBM once, into LDS (grids up to
// ks 110: 48 KB), its maximum taken there and the quotients written back -- one read and one write per pixel; larger grids
// are read twice (the second pass mostly out of L2).
// `stride`: floats between the grids of consecutive points (tile, or 2 * tile for one half of an interleaved [N, 2, ks, ks] array).
template <bool LDS_TILE>
__global__ void __launch_bounds__(kBlock) k_psf_normalize(float* __restrict__ psf, int tile, int64_t stride)
{
    extern __shared__ __attribute__((aligned(16))) float nz_tile[];
    __shared__ float red[kBlock / 64];
    float* g = psf + (int64_t)blockIdx.x * stride;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < tile; i += blockDim.x) {
        const float v = g[i];
        if (LDS_TILE) nz_tile[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_max(mx, red);
    const float den = mx + 1e-6f;
    for (int i = threadIdx.x; i < tile; i += blockDim.x) g[i] = (LDS_TILE ? nz_tile[i] : g[i]) / den;
}

// ---------------------------------------------------------------------------
// fused kernels
// ---------------------------------------------------------------------------

// The END of a launch with one workgroup per point.  The SIMDs serve the OLDEST wave first, so of the four workgroups that
// share a CU the oldest runs ahead and the youngest is left to finish alone, at 2 waves per SIMD (0.6-0.8 of the 8-wave
// rate of these kernels) -- in the middle of a launch a younger workgroup takes the freed slot, at its end nobody does:
// 0.13 ms per launch of k_psf_lr whatever its length (profiles/r05/launch_fixed_cost.txt).  Workgroups of the LAST
// generation (the last 4 x CUs blocks: `prio_from`) therefore set their waves' issue priority -- s_setprio, which ranks
// above age -- by the passes they still have to run, one level per kPrioStep passes: whoever is behind is served first
// and the workgroups of a CU finish together.  A scheduling hint only: the results are the same bit for bit.
// Measured (profiles/r06/prio_ab_levels*.txt, same box, warm): config 2 cut to 1024 / 2048 / 4096 / 16384 points
// 0.670 / 1.233 / 2.348 / 9.02 ms -> 0.596 / 1.164 / 2.275 / 8.93; one level per 1 or 3 passes, per quarter of the work, a
// constant priority for the older generations, two generations, every workgroup: all behind.  Cutting the last
// generation into smaller work units on top of it (slices of spp whose partial tiles the last slice to arrive adds up:
// commit acacd51, profiles/r06/tail_ab_*.txt) never added anything: what such a slice saves at the end of the launch it
// pays at its own start and end.
constexpr int kPrioStep = 2;
__device__ __forceinline__ void prio_by_work_left(int passes_left)
{
    const int q = (passes_left - 1) / kPrioStep;        // 3+: seven or more passes to go ... 0: the last two
    if (q >= 3) __builtin_amdgcn_s_setprio(3);
    else if (q == 2) __builtin_amdgcn_s_setprio(2);
    else if (q == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// psf_center: one workgroup per point, Sc rays, fp64 partial sums reduced in a
// fixed order (deterministic).
template <class HotMath>
__global__ void __launch_bounds__(kFused, 8)
k_chief_center(TripTable trips /* kernarg offset 0 */, const DevSurface* __restrict__ lens, int K,
               const float* __restrict__ po, const float* __restrict__ xc,
               const float* __restrict__ yc, int Sc, float pz, float zs,
               float* __restrict__ center, int32_t* __restrict__ any_valid,
               uint32_t* __restrict__ conv_mask, int prio_from)
{
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    __shared__ double red[3][kFused];
    __shared__ int red_any;
    const int n = blockIdx.x;
    const bool by_work_left = (int)blockIdx.x >= prio_from;
    const int passes = (Sc + kFused - 1) / kFused;
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    if (threadIdx.x == 0) red_any = 0;
    __syncthreads();
    const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int any = 0;
    for (int s = threadIdx.x, pass = 0; s < Sc; s += blockDim.x, ++pass) {
        if (by_work_left) prio_by_work_left(passes - pass);
        Ray r = make_ray<HotMath>(px, py, pzo, xc[s], yc[s], pz);
        trace_ray<true, HotMath>(lens, 0, K, kernarg_at(0), r, conv_mask ? lds_mask : nullptr);
        propagate_to<HotMath>(r, zs);
        sx += (double)(r.ox * r.ra);
        sy += (double)(r.oy * r.ra);
        sr += (double)r.ra;
        any |= (r.ra == 1.0f);
    }
    if (by_work_left) __builtin_amdgcn_s_setprio(0);
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

// ---------------------------------------------------------------------------
// speculate -> verify ON THE DEVICE -> re-render once (sdirt_psf_lr_verified)
// ---------------------------------------------------------------------------
// Control block of a verified call, uint32 words in device memory (zeroed by the caller):
// what the host needs to accept the result -- or to go on correcting -- in ONE readback.
constexpr int kCtlStatus = SDIRT_CTL_STATUS, kCtlAnyValid = SDIRT_CTL_ANY_VALID,
              kCtlTrips2P = SDIRT_CTL_TRIPS2, kCtlTrips2C = SDIRT_CTL_TRIPS2 + 16,
              kCtlMask1P = SDIRT_CTL_MASKS, kCtlMask1C = SDIRT_CTL_MASKS + 64,
              kCtlMask2P = SDIRT_CTL_MASKS + 128, kCtlMask2C = SDIRT_CTL_MASKS + 192;
static_assert(kCtlMask2C + 64 == SDIRT_CTL_WORDS, "control block layout");

// Arguments of the split path (several workgroups per point) of a verified call.
struct SplitArgs {
    const uint32_t* gate;        // round 2: the control block; the kernel returns at once when its
                                 // status word says that round 1 was already right
    const uint32_t* trips_dev;   // round 2: the corrected trip table (TripTable layout) in device memory
    const double* part;          // chief-ray partial sums [N][nslice][3] (sx, sy, sr), or nullptr
    int nslice;
};

// The chief-ray pass of a point cut into `nslice` workgroups (N = 64 points alone leave three
// quarters of the chip idle and every lane with four rays in a row): each writes its fp64 partial
// sums; the consumers add them in slice order (a fixed order: deterministic).  Also clears the
// output grids the primary pass will add to (global float atomics).
template <class HotMath>
__global__ void __launch_bounds__(kFused, 8)
k_chief_slices(TripTable trips /* kernarg offset 0 */, SplitArgs sa, const DevSurface* __restrict__ lens, int K,
               const float* __restrict__ po, const float* __restrict__ xc, const float* __restrict__ yc,
               int Sc, int chunk, float pz, float zs, double* __restrict__ part, uint32_t* __restrict__ any_valid,
               uint32_t* __restrict__ conv_mask, float* __restrict__ zero_a, float* __restrict__ zero_b,
               int64_t zero_n)
{
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    __shared__ double red[3][kFused];
    __shared__ int red_any;
    if (sa.gate && sa.gate[kCtlStatus] == 0u) return;              // wave-uniform
    const void* trip_words = sa.trips_dev ? (const void*)sa.trips_dev : kernarg_at(0);
    const int n = blockIdx.x / sa.nslice, j = blockIdx.x - n * sa.nslice;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (int64_t)gridDim.x * blockDim.x) {
        zero_a[i] = 0.0f;
        if (zero_b) zero_b[i] = 0.0f;
    }
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    if (threadIdx.x == 0) red_any = 0;
    __syncthreads();
    const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int any = 0;
    const int s_end = min(Sc, (j + 1) * chunk);
    for (int s = j * chunk + threadIdx.x; s < s_end; s += blockDim.x) {
        Ray r = make_ray<HotMath>(px, py, pzo, xc[s], yc[s], pz);
        trace_ray<true, HotMath>(lens, 0, K, trip_words, r, lds_mask);
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
        double* o = part + (int64_t)blockIdx.x * 3;
        o[0] = red[0][0]; o[1] = red[1][0]; o[2] = red[2][0];
        if (red_any) atomicOr(any_valid, 1u);
    }
    if ((int)threadIdx.x < K && lds_mask[threadIdx.x]) atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
}

// sdirt_amd/newton.py: verify() on the device.  `t` ran, `mask` is what it reported; writes the
// table the next round should run (t itself when it was exactly the reference's) and says whether
// t was right.  One thread; K <= 64.
__device__ bool verify_trips(const TripTable& t, const uint32_t* __restrict__ mask,
                             const DevSurface* __restrict__ lens, int K, uint32_t* __restrict__ out_words)
{
    uint32_t nw[SDIRT_MAX_SURFACES / 4];
    for (int i = 0; i < SDIRT_MAX_SURFACES / 4; ++i) nw[i] = t.w[i];
    bool failed = false;
    for (int k = 0; k < K; ++k) {
        const int sh = (k & 3) * 8;
        const int T = (int)(int8_t)(t.w[k >> 2] >> sh);
        int nv;
        if ((lens[k].h.flags & 3u) == 0u) {
            nv = 0;                                            // planes: no Newton solve
        } else {
            int j = 0;                                         // first clear bit among 1..T
            const uint32_t m = mask[k];
            for (int b = 1; b <= T; ++b)
                if (!((m >> b) & 1u)) { j = b; break; }
            if (!failed) {
                if (T >= 1 && (j == T || (j == 0 && T == SDIRT_NEWTON_MAXITER))) continue;
                failed = true;
            }
            nv = j ? j : min(max(T, 0) + 1, SDIRT_NEWTON_MAXITER);
        }
        nw[k >> 2] = (nw[k >> 2] & ~(0xffu << sh)) | ((uint32_t)(uint8_t)(int8_t)nv << sh);
    }
    for (int i = 0; i < SDIRT_MAX_SURFACES / 4; ++i) out_words[i] = failed ? nw[i] : t.w[i];
    return !failed;
}

// sdirt_psf_call's prologue as ONE launch: both pupil mappings of a psf call (u = [theta | r2 | theta_c | r2_c] ->
// xy = [x2 | y2 | xc | yc], sdirt_pupil_samples twice) and, on request, the clearing of the control block.
__global__ void __launch_bounds__(kBlock) k_pupil_pair(const float* __restrict__ u, int S, int Sc, float pr2, float pr2c,
                                                       float* __restrict__ xy, uint32_t* __restrict__ zero, int nzero)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nzero) zero[i] = 0u;
    if (i < S) pupil_point(u[i], u[S + i], pr2, xy[i], xy[S + i]);
    else if (i < S + Sc) {
        const int c = i - S;
        pupil_point(u[2 * S + c], u[2 * S + Sc + c], pr2c, xy[2 * S + c], xy[2 * S + Sc + c]);
    }
}

// The trip rule on the device for a call with one workgroup per point (sdirt_psf_call): round 1's masks stand in the
// control block; status and the tables a correction would run go beside them.  One thread.
__global__ void k_ctl_verify(uint32_t* __restrict__ ctl, const DevSurface* __restrict__ lens, int K, TripTable tp, TripTable tc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool okp = verify_trips(tp, ctl + kCtlMask1P, lens, K, ctl + kCtlTrips2P);
    const bool okc = verify_trips(tc, ctl + kCtlMask1C, lens, K, ctl + kCtlTrips2C);
    ctl[kCtlStatus] = (okp && okc) ? 0u : (1u | (okp ? 0u : 2u) | (okc ? 0u : 4u));
}

// Round 1's masks and the any-valid flag of a control block as 0 / 1 lanes -- [primary | chief][surface][bit 0..10],
// then the flag -- so that ranks can OR them with an all-reduce(MAX) (RCCL has no bitwise OR), and back.
constexpr int kLaneBits = SDIRT_NEWTON_MAXITER + 1;
constexpr int kMaskLanes = 2 * SDIRT_MAX_SURFACES * kLaneBits;       // then: any-valid, uniform sum, its complement
static_assert(kMaskLanes + 3 == SDIRT_CTL_LANES, "lane layout");
__global__ void __launch_bounds__(kBlock) k_ctl_to_lanes(const uint32_t* __restrict__ ctl, uint32_t tag, int32_t* __restrict__ lanes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SDIRT_CTL_LANES) return;
    if (i == kMaskLanes) { lanes[i] = ctl[kCtlAnyValid] != 0u; return; }
    // the caller's tag and its complement: after a MAX over ranks the two still add up to 0x3fffffff iff all ranks agreed
    if (i > kMaskLanes) { lanes[i] = (int32_t)(i == kMaskLanes + 1 ? (tag & 0x3fffffffu) : 0x3fffffffu - (tag & 0x3fffffffu)); return; }
    const int word = i / kLaneBits, bit = i - word * kLaneBits;      // word: 0..63 primary, 64..127 chief
    lanes[i] = (int32_t)((ctl[kCtlMask1P + word] >> bit) & 1u);
}
__global__ void __launch_bounds__(2 * SDIRT_MAX_SURFACES) k_ctl_from_lanes(const int32_t* __restrict__ lanes, uint32_t* __restrict__ ctl,
                                                                         const DevSurface* __restrict__ lens, int K, TripTable tp, TripTable tc)
{
    const int word = threadIdx.x;                                    // 128 threads, one mask word each
    uint32_t m = 0u;
    for (int b = 0; b < kLaneBits; ++b) m |= (lanes[word * kLaneBits + b] != 0 ? 1u : 0u) << b;
    ctl[kCtlMask1P + word] = m;
    if (word == 0) ctl[kCtlAnyValid] = lanes[kMaskLanes] != 0 ? 1u : 0u;
    if (word == 1 || word == 2) ctl[SDIRT_CTL_TAG + word - 1] = (uint32_t)lanes[kMaskLanes + word];
    __syncthreads();
    if (word == 0 && lens) {
        const bool okp = verify_trips(tp, ctl + kCtlMask1P, lens, K, ctl + kCtlTrips2P);
        const bool okc = verify_trips(tc, ctl + kCtlMask1C, lens, K, ctl + kCtlTrips2C);
        ctl[kCtlStatus] = (okp && okc) ? 0u : (1u | (okp ? 0u : 2u) | (okc ? 0u : 4u));
    }
}

// Last kernel of a round of the split path: max-normalise L (blockIdx.y 0) and R (1), and -- round 1
// -- check both trip tables against the masks the round produced: status 0 = the reference's tables,
// nothing more to do; else bit 0 set (bit 1: primary table wrong, bit 2: chief-ray table wrong) and
// the corrected tables stand in the control block for round 2, which is already enqueued.
struct FinishArgs {
    const uint32_t* gate;      // round 2: skip unless status != 0
    uint32_t* ctl;             // round 1: verify into this block (nullptr: no verification)
    const DevSurface* lens;    // surface kinds (the same in every wavelength's table)
    int K;
    TripTable tp, tc;          // the tables round 1 ran
};
__global__ void __launch_bounds__(kBlock) k_psf_finish(float* __restrict__ l, float* __restrict__ r, int tile,
                                                       int64_t pstride, int normalize, FinishArgs fa)
{
    __shared__ float red[kBlock / 64];
    if (fa.gate && fa.gate[kCtlStatus] == 0u) return;
    float* g = (blockIdx.y == 0 ? l : r) + (int64_t)blockIdx.x * pstride;
    if (normalize) {
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, g[i]);
        mx = block_max(mx, red);
        const float den = mx + 1e-6f;
        for (int i = threadIdx.x; i < tile; i += blockDim.x) g[i] = g[i] / den;
    }
    if (fa.ctl && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const bool okp = verify_trips(fa.tp, fa.ctl + kCtlMask1P, fa.lens, fa.K, fa.ctl + kCtlTrips2P);
        const bool okc = verify_trips(fa.tc, fa.ctl + kCtlMask1C, fa.lens, fa.K, fa.ctl + kCtlTrips2C);
        fa.ctl[kCtlStatus] = (okp && okc) ? 0u : (1u | (okp ? 0u : 2u) | (okc ? 0u : 4u));
    }
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
// ACC = the accumulator type of the LDS tiles.  float: ds_add_f32, which gfx950 executes lane by lane (~193 cycles
// of the CU's LDS per wave instruction whatever the addresses, profiles/r04/lds_atomic_bench.txt); double:
// ds_add_f64 (17-33 cycles), the sum rounded to fp32 once on the way out like k_forward_integral_tiles -- chosen by
// launch_psf whenever the double tiles still leave room for four workgroups per CU (L + R: ks <= 49).
// THREADS = 1024 (SDIRT_PSF_DETERMINISTIC on grids of 50 to 70 pixels): double tiles in workgroups of 16 waves, two per
// CU -- the same 8 waves per SIMD; a workgroup waits at its barriers for the slowest of 16 waves: +0.5 % (profiles/r05/ab_tiles.txt).
template <bool HAVE_R, bool BIG, class HotMath, bool CENTER, class ACC, int THREADS = kFused>
__global__ void __launch_bounds__(THREADS, BIG ? 4 : 8)
k_psf_lr(SplatBlock sb /* kernarg offset 0 */, TripSet trips /* 64 */, TripSet trips_c /* 64 + 64 W */,
         LensSet lens_set, int K, const float* __restrict__ po, const float* __restrict__ x2,
         const float* __restrict__ y2, int S, int nsplit, int chunk, float pz, float zs, int ks, int pstride, float tr,
         float tl, const float* __restrict__ center, uint32_t flags, float* __restrict__ lout,
         float* __restrict__ rout, uint32_t* __restrict__ conv_mask, CenterArgs ca, SplitArgs sa, int prio_from)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tiles_raw[];
    ACC* __restrict__ tiles = reinterpret_cast<ACC*>(tiles_raw);    // [L | R] ks*ks each
    if (!CENTER && sa.gate && sa.gate[kCtlStatus] == 0u) return;    // round 2 of a verified call, nothing to redo
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    __shared__ float red[THREADS / 64];
    __shared__ float c_sh[2];
    const int tile = ks * ks;
    ACC* tl_ = tiles;
    ACC* trr = tiles + tile;
    const int n = blockIdx.x / nsplit;
    const int j = blockIdx.x - n * nsplit;
    const int w = blockIdx.y;
    const int N = gridDim.x / nsplit;
    // the last generation of the launch issues by the passes a workgroup still has to run (prio_by_work_left)
    const bool by_work_left = (int)(blockIdx.y * gridDim.x + blockIdx.x) >= prio_from;
    const int passes_p = (min(S, (j + 1) * chunk) - j * chunk + THREADS - 1) / THREADS;
    const int passes_c = CENTER ? (ca.Sc + THREADS - 1) / THREADS : 0;
    const DevSurface* __restrict__ lens = lens_set.p[w];
    constexpr int kTripsAt = 64, kTripsCAt = 64 + 64 * SDIRT_MAX_WAVELENGTHS;
    x2 += (int64_t)w * S; y2 += (int64_t)w * S;
    if (conv_mask) conv_mask += w * SDIRT_MAX_SURFACES;
    if (CENTER) {
        ca.xc += (int64_t)w * ca.Sc; ca.yc += (int64_t)w * ca.Sc;
        ca.center_out += (int64_t)w * N * 2;
        if (ca.any_valid) ca.any_valid += w;
        if (ca.conv_mask_c) ca.conv_mask_c += w * SDIRT_MAX_SURFACES;
    } else if (center) {
        center += (int64_t)w * N * 2;
    }
    const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];

    if (CENTER) {
        // ---- chief-ray centre of this point (same arithmetic and reduction order as
        // k_chief_center; the fp64 scratch aliases the not-yet-used tile memory)
        double* redd = reinterpret_cast<double*>(tiles_raw);         // [3][THREADS]
        if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
        if (threadIdx.x == 0) c_sh[0] = 0.0f;
        __syncthreads();
        double sx = 0.0, sy = 0.0, sr = 0.0;
        int any = 0;
        for (int s = threadIdx.x, pass = 0; s < ca.Sc; s += blockDim.x, ++pass) {
            if (by_work_left) prio_by_work_left(passes_c + passes_p - pass);
            Ray r = make_ray<HotMath>(px, py, pzo, ca.xc[s], ca.yc[s], pz);
            trace_ray<true, HotMath>(ca.lens_c, 0, K, kernarg_at(kTripsCAt + 64 * w), r,
                                     ca.conv_mask_c ? lds_mask : nullptr);
            propagate_to<HotMath>(r, zs);
            sx += (double)(r.ox * r.ra);
            sy += (double)(r.oy * r.ra);
            sr += (double)r.ra;
            any |= (r.ra == 1.0f);
        }
        redd[threadIdx.x] = sx; redd[THREADS + threadIdx.x] = sy; redd[2 * THREADS + threadIdx.x] = sr;
        if (any) c_sh[0] = 1.0f;                                     // benign race: all write 1
        __syncthreads();
        for (int off = THREADS / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                redd[threadIdx.x] += redd[threadIdx.x + off];
                redd[THREADS + threadIdx.x] += redd[THREADS + threadIdx.x + off];
                redd[2 * THREADS + threadIdx.x] += redd[2 * THREADS + threadIdx.x + off];
            }
            __syncthreads();
        }
        if (ca.conv_mask_c && (int)threadIdx.x < K && lds_mask[threadIdx.x])
            atomicOr(&ca.conv_mask_c[threadIdx.x], lds_mask[threadIdx.x]);
        const float any_f = c_sh[0];
        const float den = (float)redd[2 * THREADS] + (float)1e-9;
        const float ccx = -((float)redd[0] / den), ccy = -((float)redd[THREADS] / den);
        __syncthreads();                                             // everyone has read redd / c_sh
        if (threadIdx.x == 0) {
            ca.center_out[2 * n] = ccx; ca.center_out[2 * n + 1] = ccy;
            c_sh[0] = ccx; c_sh[1] = ccy;
            if (ca.any_valid && any_f != 0.0f) atomicOr(ca.any_valid, 1);
        }
    }

    const bool from_parts = !CENTER && sa.part != nullptr;
    if (from_parts && threadIdx.x == 0) {
        // the chief-ray centre from the slices' partial sums, added in slice order (k_chief_slices);
        // the finish is k_chief_center's
        double sx = 0.0, sy = 0.0, sr = 0.0;
        for (int q = 0; q < sa.nslice; ++q) {
            const double* pq = sa.part + ((int64_t)n * sa.nslice + q) * 3;
            sx += pq[0]; sy += pq[1]; sr += pq[2];
        }
        const float den = (float)sr + (float)1e-9;
        c_sh[0] = -((float)sx / den);
        c_sh[1] = -((float)sy / den);
        if (j == 0) { ca.center_out[2 * n] = c_sh[0]; ca.center_out[2 * n + 1] = c_sh[1]; }
    }
    for (int i = threadIdx.x; i < (HAVE_R ? 2 : 1) * tile; i += blockDim.x) tiles[i] = (ACC)0;
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    __syncthreads();

    const float cx = (CENTER || from_parts) ? c_sh[0] : center[2 * n], cy = (CENTER || from_parts) ? c_sh[1] : center[2 * n + 1];
    const int s_end = min(S, (j + 1) * chunk);
    const void* kernarg = kernarg_at(0);
    const void* primary_trips = (!CENTER && sa.trips_dev) ? (const void*)sa.trips_dev : kernarg_at(kTripsAt + 64 * w);
    auto splat = [&](float sx, float sy, float dx, float dz, float ra) {
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
        if (!splat_taps<HotMath::kFused>(gm, UDiv<HotMath>::make(gm.dy_rng), UDiv<HotMath>::make(gm.dx_rng), sx, sy, cx,
                                         cy, ra, tp))
            return;
        const float x_tan = HotMath::div(-dx, dz);
        float sl, sr;
        if (BIG) dp_weights_big(dp, x_tan, sl, sr);      // separate instantiation: the rarely
        else dp_weights_small<HotMath>(dp, UDiv<HotMath>::make(dp.fmh), x_tan, sl, sr);   // used r > 0.5 branch costs registers
        atomicAdd(&tl_[tp.i_tl], (ACC)(tp.w_tl * sl));
        atomicAdd(&tl_[tp.i_tr], (ACC)(tp.w_tr * sl));
        atomicAdd(&tl_[tp.i_bl], (ACC)(tp.w_bl * sl));
        atomicAdd(&tl_[tp.i_br], (ACC)(tp.w_br * sl));
        if (HAVE_R) {
            atomicAdd(&trr[tp.i_tl], (ACC)(tp.w_tl * sr));
            atomicAdd(&trr[tp.i_tr], (ACC)(tp.w_tr * sr));
            atomicAdd(&trr[tp.i_bl], (ACC)(tp.w_bl * sr));
            atomicAdd(&trr[tp.i_br], (ACC)(tp.w_br * sr));
        }
    };
    for (int s = j * chunk + threadIdx.x, pass = 0; s < s_end; s += blockDim.x, ++pass) {
        if (by_work_left) prio_by_work_left(passes_p - pass);
        Ray r = make_ray<HotMath>(px, py, pzo, x2[s], y2[s], pz);
        trace_ray<true, HotMath>(lens, 0, K, primary_trips, r, conv_mask ? lds_mask : nullptr);
        propagate_to<HotMath>(r, zs);
        splat(r.ox, r.oy, r.dx, r.dz, r.ra);
    }
    if (by_work_left) __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    // pstride: floats from one point's grid(s) to the next -- W * tile, or 2 * tile for SDIRT_PSF_INTERLEAVED
    float* Lg = lout + (int64_t)n * pstride + w * tile;
    float* Rg = HAVE_R ? rout + (int64_t)n * pstride + w * tile : nullptr;
    // (double tiles: every sum is rounded to fp32 ONCE here; the maximum and the quotients are taken of the rounded
    // values, i.e. of exactly what k_psf_normalize would read back)
    if (nsplit == 1) {
        if (flags & SDIRT_PSF_NORMALIZE) {
            float mx = -INFINITY;
            for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, (float)tl_[i]);
            const auto div_l = UDiv<HotMath>::make(block_max(mx, red) + 1e-6f);
            auto div_r = div_l;
            if (HAVE_R) {
                mx = -INFINITY;
                for (int i = threadIdx.x; i < tile; i += blockDim.x) mx = fmaxf(mx, (float)trr[i]);
                div_r = UDiv<HotMath>::make(block_max(mx, red) + 1e-6f);
            }
            for (int i = threadIdx.x; i < tile; i += blockDim.x) {
                Lg[i] = div_l((float)tl_[i]);
                if (HAVE_R) Rg[i] = div_r((float)trr[i]);
            }
        } else {
            for (int i = threadIdx.x; i < tile; i += blockDim.x) {
                Lg[i] = (float)tl_[i];
                if (HAVE_R) Rg[i] = (float)trr[i];
            }
        }
    } else {
        for (int i = threadIdx.x; i < tile; i += blockDim.x) {
            const float a = (float)tl_[i];
            if (a != 0.0f) atomicAdd(&Lg[i], a);
            if (HAVE_R) {
                const float b = (float)trr[i];
                if (b != 0.0f) atomicAdd(&Rg[i], b);
            }
        }
    }
    if (conv_mask && (int)threadIdx.x < K && lds_mask[threadIdx.x])
        atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
}

static void launch_normalize(float* psf, int64_t N, int tile, hipStream_t st, int64_t stride = 0)
{
    const size_t lds = sizeof(float) * (size_t)tile;
    if (stride == 0) stride = tile;
    if (lds <= 48 * 1024 - 64)            // ks <= 110: inside the default dynamic-LDS allowance, two to ten workgroups per CU
        k_psf_normalize<true><<<(unsigned)N, kBlock, lds, st>>>(psf, tile, stride);
    else
        k_psf_normalize<false><<<(unsigned)N, kBlock, 0, st>>>(psf, tile, stride);
}

// first block of the last generation of a launch of `blocks` workgroups at four per CU (prio_by_work_left)
static int last_generation_from(int64_t blocks)
{
    return (int)std::max<int64_t>(0, blocks - 4ll * device_cus_or_default());
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

// How k_forward_integral_tiles is launched for N points x S samples on grids of ks x ks (ntile of them per point,
// `acc` bytes per accumulator).  false: the grids do not fit LDS.
static bool plan_forward_integral(int64_t N, int64_t S, int ks, int ntile, size_t acc, int ncu, FiLaunch& fl)
{
    const size_t lds_max = 160 * 1024 - 1024;
    fl.stride = (ntile * ks * ks) | 1;
    const size_t per_point = acc * (size_t)fl.stride;
    if (per_point > lds_max) return false;
    // one point per workgroup unless a point has fewer samples than the workgroup has threads (whole waves per
    // point: at most kFiThreads / 64 points), and never more than LDS holds
    int P = 1;
    while (P < kFiThreads / 64 && (int64_t)(kFiThreads / (2 * P)) >= S) P *= 2;
    while (P > 1 && (size_t)P * per_point > lds_max) P /= 2;
    fl.P = P;
    const int rp = kFiThreads / P;
    fl.logRp = 0;
    while ((1 << fl.logRp) < rp) ++fl.logRp;
    fl.ngroups = (int)((N + P - 1) / P);
    // few points: cut the spp axis as well, at least two passes per slice
    int64_t nsplit = 1;
    if (fl.ngroups < 2 * ncu) nsplit = std::min<int64_t>((2 * ncu + fl.ngroups - 1) / fl.ngroups, std::max<int64_t>(1, S / (2 * rp)));
    int64_t chunk = (S + nsplit - 1) / nsplit;
    chunk = std::max<int64_t>(rp, (chunk + rp - 1) / rp * rp);
    fl.chunk = chunk;
    fl.nsplit = (int)std::max<int64_t>(1, (S + chunk - 1) / chunk);
    return true;
}

int sdirt_forward_integral_plan(int64_t N, int64_t S, int32_t ks, int32_t both, int32_t n_cus, int64_t* plan)
{
    if (!plan || N < 1 || S < 1 || n_cus < 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_ks(ks, SDIRT_MAX_KS_STAGED)) return rc;
    FiLaunch fl;
    const int ntile = both ? 2 : 1;
    const bool wide = plan_forward_integral(N, S, ks, ntile, sizeof(double), n_cus, fl);
    const bool tiles = wide || plan_forward_integral(N, S, ks, ntile, sizeof(float), n_cus, fl);
    plan[0] = tiles ? (wide ? 8 : 4) : 0;
    for (int i = 1; i < 6; ++i) plan[i] = 0;
    if (!tiles) return SDIRT_OK;
    plan[1] = fl.P; plan[2] = fl.ngroups; plan[3] = fl.nsplit; plan[4] = fl.chunk;
    plan[5] = (int64_t)plan[0] * fl.P * fl.stride;
    return SDIRT_OK;
}

int sdirt_forward_integral(sdirt_rays rays, int64_t S, int64_t N, double ps, int32_t ks,
                           const float* center, const sdirt_dp_params* dp, uint32_t flags, float* l_grid,
                           float* r_grid, void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (int rc = check_ks(ks, SDIRT_MAX_KS_STAGED)) return rc;
    if (!center || !l_grid || S < 0 || N < 0 || N > (1ll << 30)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    if (N == 0) return SDIRT_OK;
    hipStream_t st = as_stream(stream);
    const size_t bytes = sizeof(float) * (size_t)N * ks * ks;
    const DevDpParams dpp = make_dp(dp);
    const bool both = r_grid != nullptr && dpp.have_r;
    const bool strict = (flags & SDIRT_PSF_STRICT_IEEE) != 0;
    FiLaunch fl;
    int ncu = 0;
    if (int rc = device_cus(&ncu)) return rc;
    const int ntile = both ? 2 : 1;
    // double accumulators when they fit (ks <= 99 for L + R), float ones up to SDIRT_MAX_KS, else the grids stay in HBM
    const bool wide = S > 0 && plan_forward_integral(N, S, ks, ntile, sizeof(double), ncu, fl);
    const bool tiles = wide || (S > 0 && plan_forward_integral(N, S, ks, ntile, sizeof(float), ncu, fl));
    // param_list=None leaves the R grid all-zero (monte_carlo.py:230-235); grids that are added to start at zero
    if (!tiles || fl.nsplit > 1) HIP_TRY(hipMemsetAsync(l_grid, 0, bytes, st));
    if (r_grid && (!tiles || fl.nsplit > 1 || !both)) HIP_TRY(hipMemsetAsync(r_grid, 0, bytes, st));
    if (S == 0) return SDIRT_OK;
    const bool normalize = (flags & SDIRT_PSF_NORMALIZE) != 0;
    if (!tiles) {
        k_forward_integral_hbm<<<grid_for(S * N, kBlock), kBlock, 0, st>>>(
            rays, S, N, make_geom(ps, ks), dpp, center, l_grid, r_grid);
        if (normalize) {
            launch_normalize(l_grid, N, ks * ks, st);
            if (both) launch_normalize(r_grid, N, ks * ks, st);
        }
        LAUNCH_CHECK();
        return SDIRT_OK;
    }
    const size_t lds_bytes = (wide ? sizeof(double) : sizeof(float)) * (size_t)fl.P * fl.stride;
    const unsigned grid = (unsigned)fl.ngroups * (unsigned)fl.nsplit;
#define SDIRT_LAUNCH_FI_M(HR, BG, AC, MM)                                                         \
    do {                                                                                          \
        if (lds_bytes > 48 * 1024)                                                                \
            if (int rc_ = allow_large_lds<&k_forward_integral_tiles<HR, BG, AC, MM>>()) return rc_; \
        k_forward_integral_tiles<HR, BG, AC, MM><<<grid, kFiThreads, lds_bytes, st>>>(            \
            rays, S, N, make_geom(ps, ks), dpp, fl, center, l_grid, both ? r_grid : nullptr,      \
            (flags & SDIRT_PSF_NORMALIZE) ? 1 : 0);                                               \
    } while (0)
#define SDIRT_LAUNCH_FI_A(HR, BG, AC)                                                             \
    do {                                                                                          \
        if (strict) SDIRT_LAUNCH_FI_M(HR, BG, AC, Ieee); else SDIRT_LAUNCH_FI_M(HR, BG, AC, Lean); \
    } while (0)
#define SDIRT_LAUNCH_FI(HR, BG)                                                                   \
    do {                                                                                          \
        if (wide) SDIRT_LAUNCH_FI_A(HR, BG, double); else SDIRT_LAUNCH_FI_A(HR, BG, float);       \
    } while (0)
    if (both) {
        if (dpp.big) SDIRT_LAUNCH_FI(true, true); else SDIRT_LAUNCH_FI(true, false);
    } else {
        if (dpp.big) SDIRT_LAUNCH_FI(false, true); else SDIRT_LAUNCH_FI(false, false);
    }
#undef SDIRT_LAUNCH_FI_A
#undef SDIRT_LAUNCH_FI_M
#undef SDIRT_LAUNCH_FI
    if (normalize && fl.nsplit > 1) {          // the partial tiles have only just been added up in HBM
        launch_normalize(l_grid, N, ks * ks, st);
        if (both) launch_normalize(r_grid, N, ks * ks, st);
    }
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_psf_normalize(float* psf, int64_t N, int32_t ks, void* stream)
{
    if (!psf || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (ks < 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "ks < 1");
    if (N == 0) return SDIRT_OK;
    launch_normalize(psf, N, ks * ks, as_stream(stream));
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
            (float)d_sensor, center, any_valid, conv_mask, last_generation_from(N));
    else
        k_chief_center<Ieee><<<(int)N, kFused, 0, as_stream(stream)>>>(
            tt, lens->dev, lens->n_surfaces, point_obj, xc, yc, (int)Sc, (float)pupil_z,
            (float)d_sensor, center, any_valid, conv_mask, last_generation_from(N));
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

constexpr size_t kWideTilesMax = 39 * 1024;


static int spp_split(int64_t N, int64_t S, int* chunk_out, int n_cus = 0)
{
    // Fill the chip: at least ~4 workgroups per CU; split the spp axis when the
    // number of points alone cannot (e.g. PSFNet training: N=64, S=20000).
    int nsplit = 1;
    const int64_t want_blocks = (int64_t)(n_cus > 0 ? n_cus : device_cus_or_default()) * 4;
    if (N < want_blocks && S > 2 * kFused) {
        nsplit = (int)((want_blocks + N - 1) / N);
        const int max_split = (int)((S + 2 * kFused - 1) / (2 * kFused));
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    }
    // slices of equal length up to a wave: a slice rounded up to whole workgroup passes (512) left the
    // last slice nearly empty and the CUs unevenly loaded (20000 spp: 13 x 1536 + 32 -> 16 x 1280, 17 % faster)
    int chunk = (int)((S + nsplit - 1) / nsplit);
    chunk = ((chunk + 63) / 64) * 64;
    nsplit = (int)((S + chunk - 1) / (chunk > 0 ? chunk : 1));
    if (nsplit < 1) nsplit = 1;
    if (chunk_out) *chunk_out = chunk;
    return nsplit;
}

// Device scratch of a verified call: the control block, then the chief-ray partial sums.
struct VerifiedRequest {
    uint32_t* ctl;     // [SDIRT_CTL_WORDS], zeroed by the caller
    double* part;      // [N][chief slices][3]
};

// The chief-ray pass of the split path in slices: enough workgroups to touch every CU, at least
// one ray per lane and slice.
static int chief_slices(int64_t N, int64_t Sc, int* chunk_out)
{
    int64_t ns = std::min<int64_t>((Sc + kFused - 1) / kFused,
                                   (device_cus_or_default() + N - 1) / std::max<int64_t>(N, 1));
    if (ns < 1) ns = 1;
    int chunk = (int)((Sc + ns - 1) / ns);
    chunk = (chunk + 63) / 64 * 64;
    if (chunk < 64) chunk = 64;
    ns = (Sc + chunk - 1) / chunk;
    if (ns < 1) ns = 1;
    if (chunk_out) *chunk_out = chunk;
    return (int)ns;
}

static int launch_psf(const sdirt_lens* const* lens, int W, const float* point_obj, int64_t N,
                      const float* x2, const float* y2, int64_t S, double pupil_z, double d_sensor,
                      double ps, int32_t ks, const float* center, const CenterRequest* cen,
                      const sdirt_dp_params* dp, const TripSet& tt, uint32_t flags, float* l_psf,
                      float* r_psf, uint32_t* conv_mask, void* stream, const VerifiedRequest* vr = nullptr)
{
    const bool have_r = r_psf != nullptr;
    const int tile = ks * ks;
    // SDIRT_PSF_INTERLEAVED: l_psf / r_psf are the two halves of ONE [N, 2, ks, ks] array
    const bool interleaved = (flags & SDIRT_PSF_INTERLEAVED) != 0;
    if (interleaved && (W != 1 || !dp || !r_psf || r_psf != l_psf + tile))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "SDIRT_PSF_INTERLEAVED: one wavelength, dp != NULL and r_psf == l_psf + ks * ks "
                                                "(the two halves of one [N, 2, ks, ks] array)");
    const int64_t pstride = interleaved ? 2 * (int64_t)tile : (int64_t)W * tile;
    if ((int64_t)W * tile > (1ll << 30)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "grids too large");
    // a multi-wavelength launch keeps one workgroup per (point, wavelength): the chief-ray pass
    // stays fused and the whole of psf_rgb is one kernel, also for the few points of a psf_map
    int chunk = ((int)S + kFused - 1) / kFused * kFused;
    const int nsplit = W > 1 ? 1 : spp_split(N, S, &chunk);
    const int K = lens[0]->n_surfaces;

    hipStream_t st = as_stream(stream);
    const bool lean = (flags & SDIRT_PSF_STRICT_IEEE) == 0;
    const bool fuse_center = cen != nullptr && nsplit == 1;
    if (vr && (!cen || nsplit == 1 || W != 1))
        return fail(SDIRT_ERR_UNSUPPORTED, "verified call: one workgroup per point here (sdirt_psf_spp_slices == 1); "
                                           "use sdirt_psf_lr_centered");
    if (cen && !fuse_center && !vr) {                // split spp axis: centre as its own launch
        if (lean)
            k_chief_center<Lean><<<(int)N, kFused, 0, st>>>(
                cen->trips_c.t[0], cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc,
                (float)pupil_z, (float)d_sensor, cen->center_out, cen->any_valid, cen->conv_mask_c, last_generation_from(N));
        else
            k_chief_center<Ieee><<<(int)N, kFused, 0, st>>>(
                cen->trips_c.t[0], cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc,
                (float)pupil_z, (float)d_sensor, cen->center_out, cen->any_valid, cen->conv_mask_c, last_generation_from(N));
        LAUNCH_CHECK();
        center = cen->center_out;
    }
    if (nsplit > 1 && !vr) {
        HIP_TRY(hipMemsetAsync(l_psf, 0, sizeof(float) * (size_t)N * tile * (interleaved ? 2 : 1), st));
        if (have_r && !interleaved) HIP_TRY(hipMemsetAsync(r_psf, 0, sizeof(float) * (size_t)N * tile, st));
    }
    const SplatGeom gm = make_geom(ps, ks);
    const DevDpParams dpp = make_dp(dp);
    const SplatBlock sblk = make_splat_block(gm, dpp);
    const dim3 grid((unsigned)(N * nsplit), (unsigned)W);
    const bool both = have_r && dpp.have_r;
    // double accumulators (ACC of k_psf_lr) whenever they leave room for four workgroups per CU, i.e. for the
    // kernel's 8 waves per SIMD: 4 x (39 KiB + 0.4 KiB of static LDS) <= 160 KiB -- L + R up to ks 49, L alone up to 70
    const size_t n_acc = (size_t)tile * (both ? 2 : 1);
    bool wide = !dpp.big && sizeof(double) * n_acc <= kWideTilesMax;
    // SDIRT_PSF_DETERMINISTIC: double tiles also where only TWO workgroups per CU have room for them (L + R up to ks 70)
    // -- those then have 1024 threads; beyond that (or with the spp axis cut: partial grids meet in global float
    // atomics) there is no order-independent sum to offer
    bool wide1024 = false;
    if ((flags & SDIRT_PSF_DETERMINISTIC) && !wide) {
        if (dpp.big || nsplit > 1 || sizeof(double) * n_acc > 2 * kWideTilesMax)
            return fail(SDIRT_ERR_UNSUPPORTED, "SDIRT_PSF_DETERMINISTIC: float64 tiles need ks <= 70 (L + R; L alone: 99), r <= 0.5 and "
                                               "one workgroup per point (sdirt_psf_spp_slices == 1)");
        wide = wide1024 = true;
    }
    if ((flags & SDIRT_PSF_DETERMINISTIC) && nsplit > 1)
        return fail(SDIRT_ERR_UNSUPPORTED, "SDIRT_PSF_DETERMINISTIC: the spp axis is cut for this batch (sdirt_psf_spp_slices > 1): "
                                           "partial grids are added with global float atomics");
    const int threads = wide1024 ? 2 * kFused : kFused;
    size_t lds_bytes = (wide ? sizeof(double) : sizeof(float)) * n_acc;
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
        lds_bytes = std::max(lds_bytes, sizeof(double) * 3 * threads);  // fp64 reduction scratch
    }
    SplitArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    // ---- verified call on the split path: round 1 with the speculated tables, the tables checked on
    // the device by the round's last kernel, round 2 with the corrected tables enqueued right behind
    // it -- its kernels return at once when round 1 was right.  No host round trip in between.
    const int rounds = (vr && !(flags & SDIRT_PSF_ONE_ROUND)) ? 2 : 1;
    int chunk_c = 0;
    const int nslice_c = vr ? chief_slices(N, cen->Sc, &chunk_c) : 0;
    for (int round = 0; round < rounds; ++round) {
    if (vr) {
        uint32_t* ctl = vr->ctl;
        sa.gate = round ? ctl : nullptr;
        sa.part = vr->part;
        sa.nslice = nslice_c;
        sa.trips_dev = round ? ctl + kCtlTrips2C : nullptr;
        uint32_t* mask_c = ctl + (round ? kCtlMask2C : kCtlMask1C);
        float* zr = (both && !interleaved) ? r_psf : nullptr;      // interleaved: one run of N * 2 * tile floats
        const int64_t zn = (int64_t)N * tile * (interleaved ? 2 : 1);
        if (lean)
            k_chief_slices<Lean><<<(int)(N * nslice_c), kFused, 0, st>>>(
                cen->trips_c.t[0], sa, cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc, chunk_c,
                (float)pupil_z, (float)d_sensor, vr->part, ctl + kCtlAnyValid, mask_c, l_psf, zr, zn);
        else
            k_chief_slices<Ieee><<<(int)(N * nslice_c), kFused, 0, st>>>(
                cen->trips_c.t[0], sa, cen->lens_c->dev, K, point_obj, cen->xc, cen->yc, (int)cen->Sc, chunk_c,
                (float)pupil_z, (float)d_sensor, vr->part, ctl + kCtlAnyValid, mask_c, l_psf, zr, zn);
        LAUNCH_CHECK();
        sa.trips_dev = round ? ctl + kCtlTrips2P : nullptr;
        conv_mask = ctl + (round ? kCtlMask2P : kCtlMask1P);
        ca.center_out = cen->center_out;
        center = nullptr;
    }
#define SDIRT_LAUNCH_PSF_T(HR, BG, MM, CT, AC, TH)                                                \
    do {                                                                                          \
        if (lds_bytes > 48 * 1024) /* large tiles: opt in to the full 160 KiB of LDS */           \
            if (int rc_ = allow_large_lds<&k_psf_lr<HR, BG, MM, CT, AC, TH>>()) return rc_;       \
        k_psf_lr<HR, BG, MM, CT, AC, TH><<<grid, TH, lds_bytes, st>>>(                            \
            sblk, tt, ttc, ls, K, point_obj, x2, y2, (int)S, nsplit, chunk, (float)pupil_z,       \
            (float)d_sensor, ks, (int)pstride, dpp.tr, dpp.tl, center, flags, l_psf,              \
            both ? r_psf : nullptr,                                                               \
            conv_mask, ca, sa, last_generation_from((int64_t)grid.x * grid.y));                   \
    } while (0)
#define SDIRT_LAUNCH_PSF(HR, BG, MM, CT, AC) SDIRT_LAUNCH_PSF_T(HR, BG, MM, CT, AC, kFused)
#define SDIRT_LAUNCH_PSF_W(HR, MM)                                                                \
    do {                                                                                          \
        if (fuse_center) SDIRT_LAUNCH_PSF_T(HR, false, MM, true, double, 2 * kFused);             \
        else SDIRT_LAUNCH_PSF_T(HR, false, MM, false, double, 2 * kFused);                        \
    } while (0)
#define SDIRT_LAUNCH_PSF_C(HR, BG, MM, AC)                                                        \
    do {                                                                                          \
        if (fuse_center) SDIRT_LAUNCH_PSF(HR, BG, MM, true, AC); else SDIRT_LAUNCH_PSF(HR, BG, MM, false, AC); \
    } while (0)
#define SDIRT_LAUNCH_PSF_M(HR, BG, AC)                                                            \
    do {                                                                                          \
        if (lean) SDIRT_LAUNCH_PSF_C(HR, BG, Lean, AC); else SDIRT_LAUNCH_PSF_C(HR, BG, Ieee, AC); \
    } while (0)
    // the corner-clipped microlens branch runs at 4 waves per SIMD whatever the tiles: float tiles only
    if (wide1024) {
        if (both) { if (lean) SDIRT_LAUNCH_PSF_W(true, Lean); else SDIRT_LAUNCH_PSF_W(true, Ieee); }
        else { if (lean) SDIRT_LAUNCH_PSF_W(false, Lean); else SDIRT_LAUNCH_PSF_W(false, Ieee); }
    } else if (both) {
        if (dpp.big) SDIRT_LAUNCH_PSF_M(true, true, float);
        else if (wide) SDIRT_LAUNCH_PSF_M(true, false, double);
        else SDIRT_LAUNCH_PSF_M(true, false, float);
    } else {
        if (dpp.big) SDIRT_LAUNCH_PSF_M(false, true, float);
        else if (wide) SDIRT_LAUNCH_PSF_M(false, false, double);
        else SDIRT_LAUNCH_PSF_M(false, false, float);
    }
    LAUNCH_CHECK();
    if (vr) {
        FinishArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.gate = round ? vr->ctl : nullptr;
        fa.ctl = round ? nullptr : vr->ctl;
        fa.lens = lens[0]->dev; fa.K = K; fa.tp = tt.t[0]; fa.tc = cen->trips_c.t[0];
        k_psf_finish<<<dim3((unsigned)N, both ? 2u : 1u), kBlock, 0, st>>>(
            l_psf, r_psf, tile, pstride, (flags & SDIRT_PSF_NORMALIZE) ? 1 : 0, fa);
        LAUNCH_CHECK();
    }
    }   // rounds
#undef SDIRT_LAUNCH_PSF_M
#undef SDIRT_LAUNCH_PSF_C
#undef SDIRT_LAUNCH_PSF_W
#undef SDIRT_LAUNCH_PSF
#undef SDIRT_LAUNCH_PSF_T
    // param_list=None leaves the R grid all-zero (monte_carlo.py:230-235)
    if (!both && have_r) HIP_TRY(hipMemsetAsync(r_psf, 0, sizeof(float) * (size_t)N * W * tile, st));
    if (!vr && nsplit > 1 && (flags & SDIRT_PSF_NORMALIZE)) {
        launch_normalize(l_psf, N, tile, st, pstride);
        if (have_r && dpp.have_r) launch_normalize(r_psf, N, tile, st, pstride);
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

int sdirt_psf_rgb(const sdirt_lens* const* lens, int32_t W, const float* point_obj, int64_t N, const float* x2,
                  const float* y2, int64_t S, double pupil_z, double d_sensor, double ps, int32_t ks,
                  const float* center, const sdirt_dp_params* dp, const int32_t* trips, uint32_t flags,
                  float* l_psf, float* r_psf, uint32_t* conv_mask, void* stream)
{
    if (!lens || W < 1 || W > SDIRT_MAX_WAVELENGTHS || !point_obj || !x2 || !y2 || !center || !l_psf || N < 0 ||
        S < 0 || S > (1ll << 30) || N > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    for (int w = 0; w < W; ++w)
        if (!lens[w] || lens[w]->n_surfaces != lens[0]->n_surfaces)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "lens[%d] missing or surface count differs from lens[0]", w);
    if (int rc = check_ks(ks)) return rc;
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    const int K = lens[0]->n_surfaces;
    TripSet tt;
    std::memset(&tt, 0, sizeof(tt));
    for (int w = 0; w < W; ++w)
        if (int rc = make_trips(lens[w], trips ? trips + (size_t)w * K : nullptr, tt.t[w])) return rc;
    if (N == 0) return SDIRT_OK;
    return launch_psf(lens, W, point_obj, N, x2, y2, S, pupil_z, d_sensor, ps, ks, center, nullptr, dp, tt, flags,
                      l_psf, r_psf, conv_mask, stream);
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

int32_t sdirt_psf_spp_slices(int64_t N, int64_t S, int32_t n_cus)
{
    if (N < 1 || S < 1) return 1;
    return spp_split(N, S, nullptr, n_cus);
}

static size_t ctl_bytes() { return (sizeof(uint32_t) * SDIRT_CTL_WORDS + 63) / 64 * 64; }

int64_t sdirt_psf_verified_scratch_bytes(int64_t N, int64_t Sc)
{
    if (N < 0 || Sc < 0) return -1;
    return (int64_t)ctl_bytes() + (int64_t)sizeof(double) * 3 * N * chief_slices(std::max<int64_t>(N, 1), Sc, nullptr);
}

int sdirt_psf_lr_verified(const sdirt_lens* lens, const sdirt_lens* lens_center, const float* point_obj,
                          int64_t N, const float* x2, const float* y2, int64_t S, const float* xc,
                          const float* yc, int64_t Sc, double pupil_z, double d_sensor, double ps, int32_t ks,
                          const sdirt_dp_params* dp, const int32_t* trips, const int32_t* trips_center,
                          uint32_t flags, float* center, float* l_psf, float* r_psf, void* scratch,
                          void* stream)
{
    if (!lens || !lens_center || !point_obj || !x2 || !y2 || !xc || !yc || !center || !l_psf || !scratch ||
        N < 0 || S < 0 || Sc < 0 || S > (1ll << 30) || Sc > (1ll << 30) || N > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (((uintptr_t)scratch) & 7) return fail(SDIRT_ERR_INVALID_ARGUMENT, "scratch must be 8-byte aligned");
    if (lens->n_surfaces != lens_center->n_surfaces)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface count of lens and lens_center differ");
    if (int rc = check_ks(ks)) return rc;
    if (dp && !(dp->r > 0.0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "dp->r must be > 0");
    if (!trips || !trips_center) return fail(SDIRT_ERR_INVALID_ARGUMENT, "a verified call needs both speculated tables");
    for (int k = 0; k < lens->n_surfaces; ++k)
        if (trips[k] < 0 || trips_center[k] < 0)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "a verified call runs the reference's batch-wide counts: no negative (per-wave) entries");
    TripSet tt;
    CenterRequest cr;
    std::memset(&tt, 0, sizeof(tt));
    std::memset(&cr.trips_c, 0, sizeof(cr.trips_c));
    if (int rc = make_trips(lens, trips, tt.t[0])) return rc;
    if (int rc = make_trips(lens_center, trips_center, cr.trips_c.t[0])) return rc;
    if (N == 0) return SDIRT_OK;
    VerifiedRequest vr;
    vr.ctl = static_cast<uint32_t*>(scratch);
    vr.part = reinterpret_cast<double*>(static_cast<char*>(scratch) + ctl_bytes());
    cr.lens_c = lens_center; cr.xc = xc; cr.yc = yc; cr.Sc = Sc; cr.center_out = center;
    cr.any_valid = nullptr; cr.conv_mask_c = nullptr;
    return launch_psf(&lens, 1, point_obj, N, x2, y2, S, pupil_z, d_sensor, ps, ks, nullptr, &cr, dp, tt, flags,
                      l_psf, r_psf, nullptr, stream, &vr);
}

static size_t align64(size_t n) { return (n + 63) / 64 * 64; }

int64_t sdirt_psf_call_scratch_bytes(int64_t N, int64_t S, int64_t Sc)
{
    if (N < 0 || S < 0 || Sc < 0) return -1;
    return (int64_t)align64((size_t)sdirt_psf_verified_scratch_bytes(N, Sc)) + (int64_t)sizeof(float) * 4 * (S + Sc);
}

int sdirt_psf_call(const sdirt_lens* lens, const sdirt_lens* lens_center, const float* point_obj, int64_t N,
                   const float* u_host, int64_t S, int64_t Sc, double pupil_r, double pupil_r_center,
                   double pupil_z, double d_sensor, double ps, int32_t ks, const sdirt_dp_params* dp,
                   const int32_t* trips, const int32_t* trips_center, uint32_t flags, float* center,
                   float* l_psf, float* r_psf, void* scratch, uint32_t* ctl_host, void* stream)
{
    if (!lens || !lens_center || !u_host || !scratch || N < 0 || S < 1 || Sc < 1 || S > (1ll << 30) || Sc > (1ll << 30))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (((uintptr_t)scratch) & 7) return fail(SDIRT_ERR_INVALID_ARGUMENT, "scratch must be 8-byte aligned");
    hipStream_t st = as_stream(stream);
    const int64_t n = 2 * (S + Sc);
    uint32_t* ctl = static_cast<uint32_t*>(scratch);
    float* u = reinterpret_cast<float*>(static_cast<char*>(scratch) +
                                        align64((size_t)sdirt_psf_verified_scratch_bytes(N, Sc)));
    float* xy = u + n;
    // The uniforms: page-locked host memory is mapped into the device's address space -- the mapping kernel reads them
    // where they are (48 KB over PCIe, a few microseconds) instead of behind a copy on the stream.  A copy command between
    // two kernels of one stream costs two engine hand-overs, compute queue -> SDMA engine -> compute queue: 12 us each, 25
    // of the 38 us a 2048-point step idled between its fused kernels (profiles/r06/step_timeline_plain.txt).  Memory that
    // is not mapped (pageable memory handed in against the contract) still goes through the copy.
    const float* u_src = nullptr;
    {
        void* mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, const_cast<float*>(u_host), 0) == hipSuccess && mapped)
            u_src = static_cast<const float*>(mapped);
        else
            (void)hipGetLastError();
    }
    if (!u_src) {
        HIP_TRY(hipMemcpyAsync(u, u_host, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, st));
        u_src = u;
    }
    // the reference's draw order (optics.py:483-484, then again inside psf_center): theta, r^2 of the primary
    // pass, theta, r^2 of the chief-ray pass -- both mappings (and SDIRT_PSF_ZERO_CTL) in one launch
    k_pupil_pair<<<grid_for(std::max<int64_t>(S + Sc, SDIRT_CTL_WORDS), kBlock, 1 << 30), kBlock, 0, st>>>(
        u_src, (int)S, (int)Sc, (float)(pupil_r * pupil_r), (float)(pupil_r_center * pupil_r_center), xy, ctl,
        (flags & SDIRT_PSF_ZERO_CTL) ? SDIRT_CTL_WORDS : 0);
    LAUNCH_CHECK();
    const uint32_t kflags = flags & ~(SDIRT_PSF_ZERO_CTL | SDIRT_PSF_NO_VERIFY);
    if (N == 0) {
        // an empty shard of a sharded batch: nothing to render, its (cleared) control block still enters the reductions
    } else if (spp_split(N, S, nullptr) > 1) {
        if (flags & SDIRT_PSF_NO_VERIFY)
            return fail(SDIRT_ERR_UNSUPPORTED, "SDIRT_PSF_NO_VERIFY: only for calls with one workgroup per point (sdirt_psf_spp_slices == 1)");
        if (int rc = sdirt_psf_lr_verified(lens, lens_center, point_obj, N, xy, xy + S, S, xy + 2 * S, xy + 2 * S + Sc, Sc,
                                           pupil_z, d_sensor, ps, ks, dp, trips, trips_center, kflags, center, l_psf, r_psf,
                                           scratch, stream))
            return rc;
    } else {
        // one workgroup per point: ONE fused launch with the speculated tables, its masks and the any-valid flag
        // straight into the control block, the trip rule evaluated behind it by one thread (status, corrected tables);
        // a correction is the caller's next call (SDIRT_PSF_ONE_ROUND is implied)
        if (!trips || !trips_center) return fail(SDIRT_ERR_INVALID_ARGUMENT, "sdirt_psf_call needs both speculated tables");
        if (int rc = sdirt_psf_lr_centered(lens, lens_center, point_obj, N, xy, xy + S, S, xy + 2 * S, xy + 2 * S + Sc, Sc,
                                           pupil_z, d_sensor, ps, ks, dp, trips, trips_center, kflags, center,
                                           reinterpret_cast<int32_t*>(ctl + kCtlAnyValid), l_psf, r_psf, ctl + kCtlMask1P,
                                           ctl + kCtlMask1C, stream))
            return rc;
        if (!(flags & SDIRT_PSF_NO_VERIFY)) {
            TripTable tp, tc;
            if (int rc = make_trips(lens, trips, tp)) return rc;
            if (int rc = make_trips(lens_center, trips_center, tc)) return rc;
            k_ctl_verify<<<1, 64, 0, st>>>(ctl, lens->dev, lens->n_surfaces, tp, tc);
            LAUNCH_CHECK();
        }
    }
    if (ctl_host)
        HIP_TRY(hipMemcpyAsync(ctl_host, scratch, sizeof(uint32_t) * SDIRT_CTL_WORDS, hipMemcpyDeviceToHost, st));
    return SDIRT_OK;
}

int sdirt_ctl_to_lanes(const uint32_t* ctl, uint32_t tag, int32_t* lanes, void* stream)
{
    if (!ctl || !lanes) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    k_ctl_to_lanes<<<grid_for(SDIRT_CTL_LANES, kBlock), kBlock, 0, as_stream(stream)>>>(ctl, tag, lanes);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_ctl_from_lanes(const int32_t* lanes, const sdirt_lens* lens, const int32_t* trips, const int32_t* trips_center,
                         uint32_t* ctl, uint32_t* ctl_host, void* stream)
{
    if (!lanes || !ctl || (lens && (!trips || !trips_center))) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    TripTable tp, tc;
    std::memset(&tp, 0, sizeof(tp));
    std::memset(&tc, 0, sizeof(tc));
    if (lens) {
        if (int rc = make_trips(lens, trips, tp)) return rc;
        if (int rc = make_trips(lens, trips_center, tc)) return rc;
    }
    hipStream_t st = as_stream(stream);
    k_ctl_from_lanes<<<1, 2 * SDIRT_MAX_SURFACES, 0, st>>>(lanes, ctl, lens ? lens->dev : nullptr, lens ? lens->n_surfaces : 0, tp, tc);
    LAUNCH_CHECK();
    if (ctl_host)
        HIP_TRY(hipMemcpyAsync(ctl_host, ctl, sizeof(uint32_t) * SDIRT_CTL_WORDS, hipMemcpyDeviceToHost, st));
    return SDIRT_OK;
}

}  // extern "C"
