// sdirt_trace.hip -- lens tables and the STAGED ray API of libsdirt_dp.so (MI355X / gfx950 only):
// object points, pupil samples, SoA rays, sequential trace, propagation, centroid -- the kernels
// behind Lensgroup.sample_from_points / trace / trace2sensor / psf_center(rays) (deeplens/optics.py:
// 460-494, 601-717, 889-904).  The fused PSF kernels live in sdirt_psf.hip, the per-pixel
// convolutions in sdirt_render.hip.  See include/sdirt_dp.h for the ABI, DESIGN.md for layouts.
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
    pupil_point(ut[s], ur[s], pr2, x2[s], y2[s]);
}

// sample_from_points (optics.py:486-494) into a point-major SoA bundle (ray (s, n) = element n S + s): a pure
// streaming WRITE (32 bytes per ray against 12 / S + 8 / N read), so the kernel is shaped for the store path.
// VEC: every thread makes FOUR consecutive samples of one point (S % 4 == 0, arrays 16-byte aligned) and stores
// each component as one dwordx4 -- a wave instruction writes 1 KiB contiguous; the (n, s) pair is divided out
// once per thread and then stepped.
template <bool VEC>
__global__ void __launch_bounds__(kBlock)
k_sample_rays(const float* __restrict__ po, int64_t N, const float* __restrict__ x2,
              const float* __restrict__ y2, int64_t S, float pz, sdirt_rays R)
{
    constexpr int W = VEC ? 4 : 1;
    const int64_t SQ = S / W, MQ = N * SQ;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= MQ) return;
    int64_t n = q / SQ, sq = q - n * SQ;
    const int64_t dn = stride / SQ, dsq = stride - dn * SQ;
    for (; q < MQ; q += stride) {
        const float px = po[3 * n], py = po[3 * n + 1], pzo = po[3 * n + 2];
        const int64_t i = n * S + sq * W;
        if (VEC) {
            const float4 xs = *reinterpret_cast<const float4*>(x2 + 4 * sq), ys = *reinterpret_cast<const float4*>(y2 + 4 * sq);
            const Ray r0 = make_ray(px, py, pzo, xs.x, ys.x, pz), r1 = make_ray(px, py, pzo, xs.y, ys.y, pz),
                      r2 = make_ray(px, py, pzo, xs.z, ys.z, pz), r3 = make_ray(px, py, pzo, xs.w, ys.w, pz);
            *reinterpret_cast<float4*>(R.ox + i) = make_float4(r0.ox, r1.ox, r2.ox, r3.ox);
            *reinterpret_cast<float4*>(R.oy + i) = make_float4(r0.oy, r1.oy, r2.oy, r3.oy);
            *reinterpret_cast<float4*>(R.oz + i) = make_float4(r0.oz, r1.oz, r2.oz, r3.oz);
            *reinterpret_cast<float4*>(R.dx + i) = make_float4(r0.dx, r1.dx, r2.dx, r3.dx);
            *reinterpret_cast<float4*>(R.dy + i) = make_float4(r0.dy, r1.dy, r2.dy, r3.dy);
            *reinterpret_cast<float4*>(R.dz + i) = make_float4(r0.dz, r1.dz, r2.dz, r3.dz);
            *reinterpret_cast<float4*>(R.ra + i) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            if (R.obliq) *reinterpret_cast<float4*>(R.obliq + i) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        } else {
            store_ray(R, i, make_ray(px, py, pzo, x2[sq], y2[sq], pz));
        }
        n += dn; sq += dsq;
        if (sq >= SQ) { sq -= SQ; n += 1; }
    }
}

// Element j of the reference's [S, N(, 3)] tensors <-> element of the bundle: the same index for a flat bundle
// (N <= 1), n S + s for a point-major one.
__device__ __forceinline__ int64_t soa_index(int64_t j, int64_t S, int64_t N)
{
    if (N <= 1) return j;
    const int64_t s = j / N, n = j - s * N;
    return n * S + s;
}

__global__ void k_rays_from_aos(const float* __restrict__ o, const float* __restrict__ d,
                                const float* __restrict__ ra, int64_t M, int64_t S, int64_t N, int normalize, sdirt_rays R)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M;
         j += (int64_t)gridDim.x * blockDim.x) {
        Ray r;
        r.ox = o[3 * j]; r.oy = o[3 * j + 1]; r.oz = o[3 * j + 2];
        r.dx = d[3 * j]; r.dy = d[3 * j + 1]; r.dz = d[3 * j + 2];
        if (normalize) normalize3<Ieee>(r.dx, r.dy, r.dz);
        r.ra = ra ? ra[j] : 1.0f;
        r.ob = 1.0f;
        store_ray(R, soa_index(j, S, N), r);
    }
}

__global__ void k_rays_to_aos(sdirt_rays R, int64_t M, int64_t S, int64_t N, float* __restrict__ o, float* __restrict__ d)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = soa_index(j, S, N);
        if (o) { o[3 * j] = R.ox[i]; o[3 * j + 1] = R.oy[i]; o[3 * j + 2] = R.oz[i]; }
        if (d) { d[3 * j] = R.dx[i]; d[3 * j + 1] = R.dy[i]; d[3 * j + 2] = R.dz[i]; }
    }
}

// R: the bundle read, W: the bundle written (the same arrays for an in-place trace: every ray is read before it is
// written, by the thread that writes it).
// TO_SENSOR: trace2sensor (optics.py:638-664) -- the traced ray is carried on to the plane z = z_sensor
// (Ray.propagate_to) before it is stored: one pass over the bundle instead of two.
template <bool FWD, class MP, bool PREFETCH, bool TO_SENSOR = false>
__global__ void __launch_bounds__(kBlock)
k_trace(TripTable trips /* kernarg offset 0 */, const DevSurface* __restrict__ lens, int K, int first,
        int last, sdirt_rays R, sdirt_rays W, int64_t M, uint32_t* __restrict__ conv_mask, float z_sensor)
{
    __shared__ uint32_t lds_mask[SDIRT_MAX_SURFACES];
    if (threadIdx.x < SDIRT_MAX_SURFACES) lds_mask[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
         i += (int64_t)gridDim.x * blockDim.x) {
        Ray r = load_ray(R, i);
        trace_ray<FWD, MP, PREFETCH>(lens, first, last, kernarg_at(0), r, conv_mask ? lds_mask : nullptr);
        if (TO_SENSOR) propagate_to<MP>(r, z_sensor);
        store_ray(W, i, r);
    }
    __syncthreads();
    if (conv_mask && (int)threadIdx.x < K && lds_mask[threadIdx.x])
        atomicOr(&conv_mask[threadIdx.x], lds_mask[threadIdx.x]);
}

// Ray.propagate_to (basics.py:256-264) on SoA rays: reads 24, writes 12 bytes per ray and divides once -- a
// streaming kernel.  VEC: four rays per thread, every component moved as dwordx4 (M % 4 == 0, aligned arrays).
template <bool VEC>
__global__ void __launch_bounds__(kBlock) k_propagate(float z, sdirt_rays R, int64_t M)
{
    if (VEC) {
        const int64_t MQ = M / 4;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < MQ; q += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = 4 * q;
            const float4 oz = *reinterpret_cast<const float4*>(R.oz + i), dz = *reinterpret_cast<const float4*>(R.dz + i);
            const float4 ox = *reinterpret_cast<const float4*>(R.ox + i), dx = *reinterpret_cast<const float4*>(R.dx + i);
            const float4 oy = *reinterpret_cast<const float4*>(R.oy + i), dy = *reinterpret_cast<const float4*>(R.dy + i);
            const float t0 = (z - oz.x) / dz.x, t1 = (z - oz.y) / dz.y, t2 = (z - oz.z) / dz.z, t3 = (z - oz.w) / dz.w;
            *reinterpret_cast<float4*>(R.ox + i) = make_float4(ox.x + dx.x * t0, ox.y + dx.y * t1, ox.z + dx.z * t2, ox.w + dx.w * t3);
            *reinterpret_cast<float4*>(R.oy + i) = make_float4(oy.x + dy.x * t0, oy.y + dy.y * t1, oy.z + dy.z * t2, oy.w + dy.w * t3);
            *reinterpret_cast<float4*>(R.oz + i) = make_float4(oz.x + dz.x * t0, oz.y + dz.y * t1, oz.z + dz.z * t2, oz.w + dz.w * t3);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
             i += (int64_t)gridDim.x * blockDim.x) {
            const float dz = R.dz[i];
            const float t = (z - R.oz[i]) / dz;
            R.ox[i] = R.ox[i] + R.dx[i] * t;
            R.oy[i] = R.oy[i] + R.dy[i] * t;
            R.oz[i] = R.oz[i] + dz * t;
        }
    }
}

// Centroid over the spp axis of a point-major bundle: one workgroup per point reads the point's contiguous run,
// fp64 partial sums per thread (samples s = t, t + 256, ...) reduced in a fixed tree order (deterministic).
__global__ void __launch_bounds__(kBlock)
k_center_from_rays(sdirt_rays R, int64_t S, int64_t N, float* __restrict__ center, int32_t* __restrict__ any_valid)
{
    __shared__ double red[3][kBlock];
    __shared__ int red_any;
    const int64_t n = blockIdx.x;
    if (threadIdx.x == 0) red_any = 0;
    __syncthreads();
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int any = 0;
    for (int64_t s = threadIdx.x; s < S; s += kBlock) {
        const int64_t i = n * S + s;
        const float ra = R.ra[i];
        sx += (double)(R.ox[i] * ra);
        sy += (double)(R.oy[i] * ra);
        sr += (double)ra;
        any |= (ra == 1.0f);
    }
    red[0][threadIdx.x] = sx; red[1][threadIdx.x] = sy; red[2][threadIdx.x] = sr;
    if (any) red_any = 1;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
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

int sdirt_sample_rays(const float* point_obj, int64_t N, const float* x2, const float* y2, int64_t S,
                      double pupil_z, sdirt_rays rays, void* stream)
{
    if (!point_obj || !x2 || !y2 || N < 0 || S < 0)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_rays(rays)) return rc;
    if (N * S == 0) return SDIRT_OK;
    // streaming kernels: ~8 workgroups per CU, the rest by grid stride (each thread keeps 128 bytes of stores in flight)
    if (S % 4 == 0 && rays_aligned16(rays) && (((uintptr_t)x2 | (uintptr_t)y2) & 15) == 0)
        k_sample_rays<true><<<grid_for(S / 4 * N, kBlock, 2048), kBlock, 0, as_stream(stream)>>>(
            point_obj, N, x2, y2, S, (float)pupil_z, rays);
    else
        k_sample_rays<false><<<grid_for(N * S, kBlock), kBlock, 0, as_stream(stream)>>>(
            point_obj, N, x2, y2, S, (float)pupil_z, rays);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_rays_from_aos(const float* o, const float* d, const float* ra, int64_t M, int64_t N, int32_t normalize,
                        sdirt_rays rays, void* stream)
{
    if (!o || !d || M < 0 || (N > 1 && M % N != 0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_rays(rays)) return rc;
    if (M == 0) return SDIRT_OK;
    k_rays_from_aos<<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>(o, d, ra, M, N > 1 ? M / N : M, N, normalize,
                                                                           rays);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_rays_to_aos(sdirt_rays rays, int64_t M, int64_t N, float* o, float* d, void* stream)
{
    if (M < 0 || (N > 1 && M % N != 0)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (int rc = check_rays(rays)) return rc;
    if (M == 0 || (!o && !d)) return SDIRT_OK;
    k_rays_to_aos<<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>(rays, M, N > 1 ? M / N : M, N, o, d);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_trace(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                const int32_t* trips, uint32_t flags, sdirt_rays rays, int64_t M,
                uint32_t* conv_mask, void* stream)
{
    return sdirt_trace_to(lens, first, last, backward, trips, flags, rays, rays, M, conv_mask, stream);
}

static int launch_trace(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                        const int32_t* trips, uint32_t flags, sdirt_rays rays, sdirt_rays out, int64_t M,
                        uint32_t* conv_mask, void* stream, bool to_sensor, double z_sensor)
{
    if (!lens) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null lens");
    if (first < 0 || last > lens->n_surfaces || first > last)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "surface range [%d,%d) outside [0,%d]", first, last,
                    lens->n_surfaces);
    if (int rc = check_rays(rays)) return rc;
    if (int rc = check_rays(out)) return rc;
    if ((rays.obliq == nullptr) != (out.obliq == nullptr))
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "obliq must be present in both bundles or in neither");
    TripTable tt;
    if (int rc = make_trips(lens, trips, tt)) return rc;
    if (M < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "n_rays < 0");
    if (M == 0) return SDIRT_OK;
    if (first == last) {                       // nothing to trace: the output bundle is the input bundle
        if (out.ox != rays.ox) {
            hipStream_t st = as_stream(stream);
            const float* src[8] = {rays.ox, rays.oy, rays.oz, rays.dx, rays.dy, rays.dz, rays.ra, rays.obliq};
            float* dst[8] = {out.ox, out.oy, out.oz, out.dx, out.dy, out.dz, out.ra, out.obliq};
            for (int c = 0; c < 8; ++c)
                if (src[c] && dst[c] != src[c])
                    HIP_TRY(hipMemcpyAsync(dst[c], src[c], sizeof(float) * (size_t)M, hipMemcpyDeviceToDevice, st));
        }
        return to_sensor ? sdirt_propagate_to(z_sensor, out, M, stream) : SDIRT_OK;
    }
    const int grid = grid_for(M, kBlock);
    const bool lean = (flags & SDIRT_PSF_STRICT_IEEE) == 0;
    const bool prefetch = (flags & SDIRT_TRACE_NO_PREFETCH) == 0;
    const float zs = (float)z_sensor;
#define SDIRT_LAUNCH_TRACE_P(FW, MM, PF, TS)                                                        \
    k_trace<FW, MM, PF, TS><<<grid, kBlock, 0, as_stream(stream)>>>(tt, lens->dev, lens->n_surfaces, \
                                                                    first, last, rays, out, M, conv_mask, zs)
#define SDIRT_LAUNCH_TRACE(FW, MM)                                                              \
    do {                                                                                        \
        if (prefetch) SDIRT_LAUNCH_TRACE_P(FW, MM, true, false); else SDIRT_LAUNCH_TRACE_P(FW, MM, false, false); \
    } while (0)
    if (to_sensor) {                           // forward, prefetching loop only (the self-test form has no use for it)
        if (backward || !prefetch) return fail(SDIRT_ERR_UNSUPPORTED, "trace2sensor traces forward with the prefetching loop");
        if (lean) SDIRT_LAUNCH_TRACE_P(true, Lean, true, true); else SDIRT_LAUNCH_TRACE_P(true, Ieee, true, true);
    } else if (backward) {
        if (lean) SDIRT_LAUNCH_TRACE(false, Lean); else SDIRT_LAUNCH_TRACE(false, Ieee);
    } else {
        if (lean) SDIRT_LAUNCH_TRACE(true, Lean); else SDIRT_LAUNCH_TRACE(true, Ieee);
    }
#undef SDIRT_LAUNCH_TRACE_P
#undef SDIRT_LAUNCH_TRACE
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_trace_to(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                   const int32_t* trips, uint32_t flags, sdirt_rays rays, sdirt_rays out, int64_t M,
                   uint32_t* conv_mask, void* stream)
{
    return launch_trace(lens, first, last, backward, trips, flags, rays, out, M, conv_mask, stream, false, 0.0);
}

int sdirt_trace2sensor(const sdirt_lens* lens, const int32_t* trips, uint32_t flags, double d_sensor, sdirt_rays rays,
                       sdirt_rays out, int64_t M, uint32_t* conv_mask, void* stream)
{
    if (!lens) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null lens");
    return launch_trace(lens, 0, lens->n_surfaces, 0, trips, flags, rays, out, M, conv_mask, stream, true, d_sensor);
}

int sdirt_propagate_to(double z, sdirt_rays rays, int64_t M, void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (M <= 0) return M < 0 ? fail(SDIRT_ERR_INVALID_ARGUMENT, "n_rays < 0") : SDIRT_OK;
    if (M % 4 == 0 && rays_aligned16(rays))
        k_propagate<true><<<grid_for(M / 4, kBlock, 2048), kBlock, 0, as_stream(stream)>>>((float)z, rays, M);
    else
        k_propagate<false><<<grid_for(M, kBlock), kBlock, 0, as_stream(stream)>>>((float)z, rays, M);
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_center_from_rays(sdirt_rays rays, int64_t S, int64_t N, float* center, int32_t* any_valid,
                           void* stream)
{
    if (int rc = check_rays(rays)) return rc;
    if (!center || S < 0 || N < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (N == 0) return SDIRT_OK;
    if (N > (1ll << 30)) return fail(SDIRT_ERR_INVALID_ARGUMENT, "n_points too large");
    k_center_from_rays<<<(unsigned)N, kBlock, 0, as_stream(stream)>>>(rays, S, N, center, any_valid);
    LAUNCH_CHECK();
    return SDIRT_OK;
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

}  // extern "C"
