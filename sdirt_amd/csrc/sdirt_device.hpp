// sdirt_device.hpp -- per-ray device math of the dual-pixel PSF path (gfx950).
//
// Everything here is scalar-per-ray fp32 arithmetic written in the SAME
// evaluation order as the reference's torch CPU ops, and the translation unit
// is compiled with -ffp-contract=off and hipcc's default correctly rounded
// fp32 divide/sqrt, so that a ray traced here is bit-identical to the same ray
// traced by an IEEE-754 CPU evaluation of the reference op sequence
// (deeplens/surfaces.py:391-830, deeplens/monte_carlo.py:135-372).
//
// Wave-uniform data (the per-surface constant block, DP-sensor parameters) is
// read through `const __restrict__` kernel-argument pointers with wave-uniform
// indices, i.e. through the scalar cache into SGPRs: it costs no VGPRs and no
// LDS bandwidth.  Per-ray state lives in VGPRs for the whole trace.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdirt {

constexpr int kMaxAi = 8;
constexpr float kNewtonStepBound = 5.0f;   // deeplens/surfaces.py:29

// Flat per-surface constant block.  Every field that the reference obtains by
// rounding a Python/numpy float64 to fp32 at the point of use is rounded on the
// host, once, in sdirt_lens_create (see sdirt_dp.hip: make_dev_surface).
struct DevSurface {
    int32_t kind;        // 0 plane, 1 sphere, 2 asphere
    int32_t ai_degree;
    int32_t do_refract;  // plane: eta != 1 (surfaces.py:450); curved: always 1
    int32_t k_gt_m1;     // k > -1 (surfaces.py:727,738)
    float d, c, k;
    float r_lim;         // fp32(r)                         surfaces.py:421
    float r2_lim;        // fp32(r*r in double)             surfaces.py:464,728
    float c2;            // c*c (fp32)
    float onepk;         // 1 + k
    float lim_loose;     // ((1/c2) * fp32(1-1e-9)) / (1+k) surfaces.py:728,739
    float d_plus_R;      // d + 1/c                         surfaces.py:607-615
    float eta_f, eta2_f; // forward: fp32(n1/n2), fp32((n1/n2)^2)   surfaces.py:401,663-669
    float eta_b, eta2_b; // backward: fp32(n2/n1), fp32((n2/n1)^2)  surfaces.py:404
    float ai[kMaxAi];    // ai2, ai4, ...
    float kai[kMaxAi];   // (i+1) * ai[i]  (python int * fp32 tensor)  surfaces.py:823
};

struct DevDpParams {
    float h, f, w, r;    // fp32 of the python floats        monte_carlo.py:157-164
    float fmh;           // fp32(f - h) evaluated in double
    float rr;            // r * r (fp32)
    float tr, tl;        // big-r only: asin(0.5/r), pi - tr  monte_carlo.py:275-276
    int32_t big;         // r > 0.5                           monte_carlo.py:59
    int32_t have_r;      // param_list is not None -> R grid is filled (:231)
    int32_t r_pow2;      // r is a power of two: x / r == x * inv_r exactly
    float inv_r;
};

// ---------------------------------------------------------------------------
// Lane types.  The trace core is written once, generic in the per-lane value type:
//   float : one ray per lane -- what every kernel instantiates;
//   f2    : two rays per lane.  Kept as a tested experiment: on gfx950 the compiler
//           turns the adds/multiplies into v_pk_mul_f32 / v_pk_add_f32, results are
//           bit-identical, but a packed fp32 op occupies the SIMD-32 for twice the
//           passes of a scalar one (the 157 TFLOP/s vector-fp32 peak is already the
//           un-packed FMA rate), so it buys nothing: measured 5.35 / 12.09 ms
//           (f2) against 5.08 / 11.76 ms (float) for k_chief_center / k_psf_lr.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <class T> struct Lane;

template <> struct Lane<float> {
    using Mask = bool;
    static constexpr int width = 1;
    static __device__ __forceinline__ float splat(float v) { return v; }
    static __device__ __forceinline__ Mask lt(float a, float b) { return a < b; }
    static __device__ __forceinline__ Mask le(float a, float b) { return a <= b; }
    static __device__ __forceinline__ Mask gt(float a, float b) { return a > b; }
    static __device__ __forceinline__ Mask ge(float a, float b) { return a >= b; }
    static __device__ __forceinline__ Mask isnan(float a) { return a != a; }
    static __device__ __forceinline__ Mask mand(Mask a, Mask b) { return a && b; }
    static __device__ __forceinline__ Mask mnot(Mask a) { return !a; }
    static __device__ __forceinline__ Mask all(bool u) { return u; }
    static __device__ __forceinline__ bool any(Mask m) { return m; }
    static __device__ __forceinline__ float sel(Mask m, float a, float b) { return m ? a : b; }
    static __device__ __forceinline__ float to01(Mask m) { return m ? 1.0f : 0.0f; }
    static __device__ __forceinline__ float fabs(float a) { return __builtin_fabsf(a); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float sqrt_ieee(float a) { return __builtin_sqrtf(a); }
    static __device__ __forceinline__ float sqrt_fast(float a) { return __builtin_amdgcn_sqrtf(a); }
    static __device__ __forceinline__ float rcp_fast(float a) { return __builtin_amdgcn_rcpf(a); }
    static __device__ __forceinline__ float rsq_fast(float a) { return __builtin_amdgcn_rsqf(a); }
    static __device__ __forceinline__ float next_up(float a) { return __uint_as_float(__float_as_uint(a) + 1u); }
    static __device__ __forceinline__ float next_down(float a) { return __uint_as_float(__float_as_uint(a) - 1u); }
    static __device__ __forceinline__ Mask is_zero_or_pinf(float a)
    {
        return __builtin_amdgcn_classf(a, 0x260);   // -0 | +0 | +inf
    }
    template <class F> static __device__ __forceinline__ float map(float a, F f) { return f(a); }
};

template <> struct Lane<f2> {
    using Mask = i2;
    static constexpr int width = 2;
    static __device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
    static __device__ __forceinline__ Mask lt(f2 a, f2 b) { return a < b; }
    static __device__ __forceinline__ Mask le(f2 a, f2 b) { return a <= b; }
    static __device__ __forceinline__ Mask gt(f2 a, f2 b) { return a > b; }
    static __device__ __forceinline__ Mask ge(f2 a, f2 b) { return a >= b; }
    static __device__ __forceinline__ Mask isnan(f2 a) { return a != a; }
    static __device__ __forceinline__ Mask mand(Mask a, Mask b) { return a & b; }
    static __device__ __forceinline__ Mask mnot(Mask a) { return ~a; }
    static __device__ __forceinline__ Mask all(bool u) { return u ? i2{-1, -1} : i2{0, 0}; }
    static __device__ __forceinline__ bool any(Mask m) { return (m.x | m.y) != 0; }
    static __device__ __forceinline__ f2 sel(Mask m, f2 a, f2 b)
    {
        return f2{m.x ? a.x : b.x, m.y ? a.y : b.y};
    }
    static __device__ __forceinline__ f2 to01(Mask m) { return f2{m.x ? 1.0f : 0.0f, m.y ? 1.0f : 0.0f}; }
    static __device__ __forceinline__ f2 fabs(f2 a) { return f2{__builtin_fabsf(a.x), __builtin_fabsf(a.y)}; }
    static __device__ __forceinline__ f2 fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ f2 sqrt_ieee(f2 a) { return f2{__builtin_sqrtf(a.x), __builtin_sqrtf(a.y)}; }
    static __device__ __forceinline__ f2 sqrt_fast(f2 a)
    {
        return f2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)};
    }
    static __device__ __forceinline__ f2 rcp_fast(f2 a)
    {
        return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
    }
    static __device__ __forceinline__ Mask is_zero_or_pinf(f2 a)
    {
        return i2{__builtin_amdgcn_classf(a.x, 0x260) ? -1 : 0, __builtin_amdgcn_classf(a.y, 0x260) ? -1 : 0};
    }
    static __device__ __forceinline__ f2 next_up(f2 a)
    {
        return f2{__uint_as_float(__float_as_uint(a.x) + 1u), __uint_as_float(__float_as_uint(a.y) + 1u)};
    }
    static __device__ __forceinline__ f2 next_down(f2 a)
    {
        return f2{__uint_as_float(__float_as_uint(a.x) - 1u), __uint_as_float(__float_as_uint(a.y) - 1u)};
    }
    static __device__ __forceinline__ f2 rsq_fast(f2 a)
    {
        return f2{__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)};
    }
    template <class F> static __device__ __forceinline__ f2 map(f2 a, F f) { return f2{f(a.x), f(a.y)}; }
};

template <class T>
struct RayT {
    T ox, oy, oz, dx, dy, dz, ra, ob;
};
using Ray = RayT<float>;

// Math policies.
//  Ieee: correctly rounded / and sqrt (hipcc's default expansion: 12- and 17-instruction
//        sequences with range scaling and special-case fix-up): bit-identical to an IEEE CPU
//        evaluation of the reference's operation sequence for ALL operands.
//  Lean: the same correctly rounded results for every NORMAL-RANGE operand, without the
//        range scaling (v_div_scale / 2^32 pre-scaling) and special-value fix-up
//        (v_div_fixup, class tests) that the compiler's sequences carry for denormal, huge
//        and zero/inf operands -- 6 and 10 instructions instead of 12 and 17:
//          div : y0 = rcp(b); y = y0 + y0*(1 - b*y0); q0 = a*y; q = q0 + y*(a - b*q0)
//                (all via fma).  Verified bit-identical to IEEE on ALL 2^46 mantissa pairs
//                (sdirt_selftest_math mode 1; the recurrence is exact-scaling in the exponent,
//                so this covers every operand pair whose exponents stay within +-60).
//                a == 0 gives the correctly signed zero; b == 0 gives NaN instead of inf.
//          sqrt: s = v_sqrt_f32(x) (<= 1 ulp), then pick among {s-ulp, s, s+ulp} by the sign
//                of the exact residuals x - s'*s (fma) -- the compiler's own correction
//                step.  Verified on EVERY positive normal fp32 (mode 0, exhaustive); +-0,
//                +inf, negative and NaN inputs behave as IEEE; denormal inputs do not.
//        No operand on a valid ray is denormal or zero-denominator (eps = 1e-9 guards, unit
//        direction vectors, |positions| in [1e-6, 2e4] mm), so valid rays are bit-identical
//        to the Ieee instantiation; tests/test_gpu_parity.py runs both against the oracle.
struct Ieee {
    template <class T> static __device__ __forceinline__ T div(T a, T b) { return a / b; }
    template <class T> static __device__ __forceinline__ T sqrt(T x) { return Lane<T>::sqrt_ieee(x); }
    template <class T> static __device__ __forceinline__ void div3(T& a0, T& a1, T& a2, T b)
    {
        a0 = a0 / b; a1 = a1 / b; a2 = a2 / b;
    }
};
struct Lean {
    template <class T> static __device__ __forceinline__ T div(T a, T b)
    {
        using L = Lane<T>;
        const T y0 = L::rcp_fast(b);
        const T y = L::fma(L::fma(-b, y0, L::splat(1.0f)), y0, y0);
        const T q0 = a * y;
        return L::fma(L::fma(-b, q0, a), y, q0);
    }
    // three numerators over one denominator: the refined reciprocal is computed once; each
    // quotient is the same arithmetic as div(), hence the same bits
    template <class T> static __device__ __forceinline__ void div3(T& a0, T& a1, T& a2, T b)
    {
        using L = Lane<T>;
        const T y0 = L::rcp_fast(b);
        const T y = L::fma(L::fma(-b, y0, L::splat(1.0f)), y0, y0);
        T q = a0 * y; a0 = L::fma(L::fma(-b, q, a0), y, q);
        q = a1 * y;   a1 = L::fma(L::fma(-b, q, a1), y, q);
        q = a2 * y;   a2 = L::fma(L::fma(-b, q, a2), y, q);
    }
    template <class T> static __device__ __forceinline__ T sqrt(T x)
    {
        using L = Lane<T>;
        T s = L::sqrt_fast(x);
        const T sm = L::next_down(s), sp = L::next_up(s);
        const T rm = L::fma(-sm, s, x);          // x - (s - ulp) * s
        const T rp = L::fma(-sp, s, x);          // x - (s + ulp) * s
        s = L::sel(L::le(rm, L::splat(0.0f)), sm, s);
        s = L::sel(L::gt(rp, L::splat(0.0f)), sp, s);
        // +-0 and +inf map to themselves; negative / NaN inputs give NaN through v_sqrt_f32
        return L::sel(L::is_zero_or_pinf(x), x, s);
    }
};
__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    // torch.clamp semantics: NaN propagates
    if (v != v) return v;
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

template <class T>
__device__ __forceinline__ T clampv(T v, float lo, float hi)
{
    using L = Lane<T>;
    // a NaN fails both comparisons and passes through, as torch.clamp does
    T w = L::sel(L::lt(v, L::splat(lo)), L::splat(lo), v);
    return L::sel(L::gt(w, L::splat(hi)), L::splat(hi), w);
}

// torch.nn.functional.normalize over a last dim of 3 (basics.py:245,
// surfaces.py:628): v / max(||v||, 1e-12); torch's CPU kernel accumulates the
// squares with fused multiply-adds (x*x, then fma y, then fma z).
template <class M, class T>
__device__ __forceinline__ void normalize3(T& x, T& y, T& z)
{
    using L = Lane<T>;
    T acc = x * x;
    acc = L::fma(y, y, acc);
    acc = L::fma(z, z, acc);
    T nrm = M::sqrt(acc);
    nrm = L::sel(L::lt(nrm, L::splat(1e-12f)), L::splat(1e-12f), nrm);
    M::div3(x, y, z, nrm);
}

// r2 ** n as torch evaluates it on CPU: n==2 -> x*x, n==3 -> (x*x)*x, n>=4 a
// <=1 ulp vector pow; for n>=4 we return the correctly rounded exact power
// (product in fp64, rounded once).
__device__ __forceinline__ float powi(float x, int n)
{
    if (n == 1) return x;
    if (n == 2) return x * x;
    if (n == 3) return (x * x) * x;
    double p = (double)x, acc = p;
    for (int i = 1; i < n; ++i) acc *= p;
    return (float)acc;
}

// surfaces.py:787-808 and 811-830 evaluated together (they share sqrt(1-a)).
template <class M, class T>
__device__ __forceinline__ void sag_g_dgd(const DevSurface& s, T r2, T& g, T& dgd)
{
    using L = Lane<T>;
    const T a = (s.onepk * r2) * s.c2;
    const T sf = M::sqrt(1.0f - a);
    const T onesf = 1.0f + sf;
    g = M::div(r2 * s.c, onesf);
    dgd = M::div((onesf + M::div(a * 0.5f, sf)) * s.c, onesf * onesf);   // a/2 == a*0.5 exactly
    if (s.ai_degree > 0) {
        dgd = dgd + s.ai[0];
        g = g + s.ai[0] * r2;
        T pw = r2;   // r2 ** i
        for (int i = 1; i < s.ai_degree; ++i) {
            dgd = dgd + s.kai[i] * pw;
            const int n = i + 1;
            pw = L::map(r2, [n](float v) { return powi(v, n); });
            g = g + s.ai[i] * pw;
        }
    }
}

template <class M, class T>
__device__ __forceinline__ T sag_dgd_only(const DevSurface& s, T r2)
{
    T g, dgd;
    sag_g_dgd<M>(s, r2, g, dgd);
    return dgd;
}

// surfaces.py:523-586.  `trips` loop iterations (wave-uniform), then the extra
// differentiable step and the validity test.  Returns Newton's own validity;
// mask_out (wave-uniform, lives in SGPRs) gets bit j set when ANY active lane of
// the wave had |f(t)| > 50e-6 in trip j -- the per-wave share of the reference's
// batch-wide `.any()` loop condition (surfaces.py:547).
template <class M, bool KGT, class T>
__device__ __forceinline__ typename Lane<T>::Mask newton_k(const DevSurface& s, const RayT<T>& r,
                                                           int trips, T& t_out, uint32_t& mask_out)
{
    using L = Lane<T>;
    using Mk = typename L::Mask;
    const float tol_loose = (float)50e-6, tol_tight = (float)10e-6, eps = (float)1e-9;
    const T t0 = M::div(s.d - r.oz, r.dz);
    const T dd = r.dx * r.dx + r.dy * r.dy;
    const T dox = r.dx * r.ox + r.dy * r.oy;
    const Mk alive = L::gt(r.ra, L::splat(0.0f));
    T t = t0;
    uint32_t mask = 0;
    // trips >= 0: exactly that many trips (the reference's batch-wide count, supplied by the host).
    // trips < 0: at most -trips, and this WAVE stops as soon as none of its 64 rays is open -- the
    // reference's loop condition evaluated per wave instead of per batch (speed mode, no host check).
    const bool adaptive = trips < 0;
    const int cap = adaptive ? -trips : trips;
    for (int it = 1; it <= cap; ++it) {
        const T nx = r.ox + r.dx * t, ny = r.oy + r.dy * t, nz = r.oz + r.dz * t;
        const T rr = nx * nx + ny * ny;
        const Mk inside = KGT ? L::lt(rr, L::splat(s.lim_loose)) : L::gt(rr, L::splat(0.0f));
        // g(x*valid, y*valid) (surfaces.py:694-696): (x*1)^2 + (y*1)^2 is rr bit for bit, and
        // (x*0)^2 + (y*0)^2 is 0 for every finite position
        const T r2 = L::sel(L::mand(inside, alive), rr, L::splat(0.0f));
        T g, dgd;
        sag_g_dgd<M>(s, r2, g, dgd);
        const T ft = (g + s.d) - nz;
        const T dr2dt = 2.0f * (dd * t + dox);
        const T dfdt = dgd * dr2dt - r.dz;
        const bool open = L::any(L::gt(L::fabs(ft), L::splat(tol_loose)));
        const bool wave_open = __ballot(open) != 0ull;
        mask |= wave_open ? (1u << it) : 0u;
        t = t - clampv(M::div(ft, dfdt + eps), -kNewtonStepBound, kNewtonStepBound);
        if (adaptive && !wave_open) break;
    }
    mask_out = mask;
    const T t1 = t - t0;   // :563
    t = t0 + t1;           // :567
    T nx = r.ox + r.dx * t, ny = r.oy + r.dy * t;
    const T nz = r.oz + r.dz * t;
    T rr = nx * nx + ny * ny;
    Mk v = L::mand(L::lt(rr, L::splat(s.r2_lim)), alive);
    if (KGT) v = L::mand(v, L::lt(rr, L::splat(s.lim_loose)));
    const T r2 = L::sel(v, rr, L::splat(0.0f));
    T g, dgd;
    sag_g_dgd<M>(s, r2, g, dgd);
    const T ft = (g + s.d) - nz;
    const T dr2dt = 2.0f * (dd * t + dox);
    const T dfdt = dgd * dr2dt - r.dz;
    t = t - clampv(M::div(ft, dfdt + eps), -kNewtonStepBound, kNewtonStepBound);
    nx = r.ox + r.dx * t;
    ny = r.oy + r.dy * t;
    rr = nx * nx + ny * ny;
    v = L::mand(L::lt(rr, L::splat(s.r2_lim)), alive);
    if (KGT) v = L::mand(v, L::lt(rr, L::splat(s.lim_loose)));
    v = L::mand(v, L::lt(L::fabs(ft), L::splat(tol_tight)));
    v = L::mand(v, L::gt(t, L::splat(0.0f)));
    t_out = t;
    return v;
}

// k > -1 (every sphere, ellipsoid, mild asphere) and k <= -1 differ only in the domain test
// (surfaces.py:727-743); the branch is wave-uniform, so it is taken once, outside the loop.
template <class M, class T>
__device__ __forceinline__ typename Lane<T>::Mask newton(const DevSurface& s, const RayT<T>& r,
                                                         int trips, T& t_out, uint32_t& mask_out)
{
    if (s.k_gt_m1) return newton_k<M, true>(s, r, trips, t_out, mask_out);
    return newton_k<M, false>(s, r, trips, t_out, mask_out);
}

// surfaces.py:633-679 with _normal (:589-630).  FWD: rays travel +z (n negated,
// eta = n1/n2); !FWD: backward tracing.
template <bool FWD, class M, class T>
__device__ __forceinline__ void refract(const DevSurface& s, RayT<T>& r)
{
    using L = Lane<T>;
    using Mk = typename L::Mask;
    T nx, ny, nz;
    if (s.kind == 0) {
        nx = L::splat(0.0f); ny = L::splat(0.0f); nz = L::splat(-1.0f);
    } else if (s.kind == 1) {
        if (s.c > 0.0f) {
            nx = 2.0f * r.ox; ny = 2.0f * r.oy; nz = 2.0f * r.oz - 2.0f * s.d_plus_R;
        } else {
            nx = -2.0f * r.ox; ny = -2.0f * r.oy; nz = -2.0f * r.oz + 2.0f * s.d_plus_R;
        }
    } else {
        const T vf = L::to01(L::gt(r.ra, L::splat(0.0f)));
        const T xv = r.ox * vf, yv = r.oy * vf;
        const T ds = sag_dgd_only<M>(s, xv * xv + yv * yv);
        nx = (ds * 2.0f) * xv; ny = (ds * 2.0f) * yv; nz = L::splat(-1.0f);
    }
    normalize3<M>(nx, ny, nz);
    if (FWD) { nx = -nx; ny = -ny; nz = -nz; }
    const float eta = FWD ? s.eta_f : s.eta_b;
    const float eta2 = FWD ? s.eta2_f : s.eta2_b;
    const T cosi = (r.dx * nx + r.dy * ny) + r.dz * nz;
    const T c2i = cosi * cosi;
    const T omc = 1.0f - c2i;
    Mk v = L::mand(L::gt(c2i, L::splat(0.1f)), L::lt(eta2 * omc, L::splat(1.0f)));
    v = L::mand(v, L::gt(r.ra, L::splat(0.0f)));
    const T vf = L::to01(v);
    const T sr = M::sqrt(1.0f - (eta2 * omc) * vf);
    T ndx = sr * nx + eta * (r.dx - cosi * nx);
    T ndy = sr * ny + eta * (r.dy - cosi * ny);
    T ndz = sr * nz + eta * (r.dz - cosi * nz);
    ndx = L::sel(v, ndx, r.dx); ndy = L::sel(v, ndy, r.dy); ndz = L::sel(v, ndz, r.dz);
    r.ob = r.ob * ((ndx * r.dx + ndy * r.dy) + ndz * r.dz);
    r.dx = ndx; r.dy = ndy; r.dz = ndz;
    r.ra = r.ra * vf;
}

// Aspheric.ray_reaction, surfaces.py:391-520.  Returns the Newton convergence
// mask of this ray on this surface (0 for planes).
template <bool FWD, class M, class T>
__device__ __forceinline__ uint32_t surface_reaction(const DevSurface& s, RayT<T>& r, int trips)
{
    using L = Lane<T>;
    using Mk = typename L::Mask;
    uint32_t mask = 0;
    if (s.kind == 0) {
        const T t = M::div(s.d - r.oz, r.dz);
        const T nx = r.ox + t * r.dx, ny = r.oy + t * r.dy, nz = r.oz + t * r.dz;
        const Mk v = L::mand(L::le(M::sqrt(nx * nx + ny * ny), L::splat(s.r_lim)),
                             L::gt(r.ra, L::splat(0.0f)));
        r.ox = L::sel(v, nx, r.ox); r.oy = L::sel(v, ny, r.oy); r.oz = L::sel(v, nz, r.oz);
        r.ra = r.ra * L::to01(v);
        if (s.do_refract) refract<FWD, M>(s, r);
        return 0;
    }
    T t;
    const Mk vn = newton<M>(s, r, trips, t, mask);
    const T nx = r.ox + t * r.dx, ny = r.oy + t * r.dy, nz = r.oz + t * r.dz;
    Mk v;
    if (s.kind == 1) {                                                                    // :464
        v = L::mand(L::le(nx * nx + ny * ny, L::splat(s.r2_lim)), L::ge(t, L::splat(0.0f)));
        v = L::mand(v, L::gt(r.ra, L::splat(0.0f)));
    } else {
        v = vn;                                                                           // :495
    }
    r.ox = L::sel(v, nx, r.ox); r.oy = L::sel(v, ny, r.oy); r.oz = L::sel(v, nz, r.oz);
    r.ra = r.ra * L::to01(v);
    refract<FWD, M>(s, r);
    return mask;
}

// Ray.propagate_to, basics.py:256-264
template <class M, class T>
__device__ __forceinline__ void propagate_to(RayT<T>& r, float z)
{
    const T t = M::div(z - r.oz, r.dz);
    r.ox = r.ox + r.dx * t;
    r.oy = r.oy + r.dy * t;
    r.oz = r.oz + r.dz * t;
}

// ---- dual-pixel closed-form sub-pixel weights -------------------------------
__device__ __forceinline__ float seg(float u)   // u - 1/2*sin(2u), monte_carlo.py:182
{
    return u - 0.5f * __ocml_sin_f32(2.0f * u);
}

// seg(acos(z)) = acos(z) - z*sqrt(1-z^2), the area function of monte_carlo.py:179-183
// with sin(2 acos z) = 2 z sqrt(1-z^2) applied.  (1-z)(1+z) instead of 1-z*z keeps the
// root accurate next to the clamped ends z = +-1 (where it is exactly 0).  The result
// differs from "sin(2*acos)" evaluated in fp32 by a few 1e-8 absolute -- less than the
// reference's own MKL sin/acos differ from libm -- and it only scales splat WEIGHTS.
__device__ __forceinline__ float seg_acos(float z)
{
    const float root = __builtin_amdgcn_sqrtf((1.0f - z) * (1.0f + z));
    return __ocml_acos_f32(z) - z * root;
}

// x / r for the microlens radius: exact multiply when r is a power of two (default 0.5)
__device__ __forceinline__ float over_r(const DevDpParams& p, float x)
{
    return p.r_pow2 ? x * p.inv_r : x / p.r;
}

// monte_carlo.py:169-206 (r <= 0.5)
__device__ __forceinline__ void dp_weights_small(const DevDpParams& p, float x_tan, float& sl,
                                                 float& sr)
{
    const float r = p.r, rr = p.rr, fmh = p.fmh;
    const float fx = p.f * x_tan;
    float xr = p.w - ((fx - p.w) * p.h) / fmh;
    float xm = ((-fx) * p.h) / fmh;
    float xl = (-p.w) - ((fx + p.w) * p.h) / fmh;
    xr = clampf(xr, -r, r); xm = clampf(xm, -r, r); xl = clampf(xl, -r, r);
    float sm = seg_acos(over_r(p, xm));
    const float sr_ml = rr * (sm - seg_acos(over_r(p, xr)));
    const float sl_ml = rr * (seg_acos(over_r(p, xl)) - sm);
    const float hx = p.h * x_tan;
    xr = p.w - hx; xm = 0.0f - hx; xl = (-p.w) - hx;
    xr = clampf(xr, -0.5f, 0.5f); xm = clampf(xm, -0.5f, 0.5f); xl = clampf(xl, -0.5f, 0.5f);
    const float xri = clampf(xr, -r, r), xmi = clampf(xm, -r, r), xli = clampf(xl, -r, r);
    sm = seg_acos(over_r(p, xmi));
    const float sr_in = rr * (sm - seg_acos(over_r(p, xri)));
    const float sl_in = rr * (seg_acos(over_r(p, xli)) - sm);
    const float sr_mg = (xr - xm) * 1.0f - sr_in;
    const float sl_mg = (xm - xl) * 1.0f - sl_in;
    sr = sr_ml + sr_mg;
    sl = sl_ml + sl_mg;
}

// monte_carlo.py:274-338 (r > 0.5)
__device__ __forceinline__ void dp_weights_big(const DevDpParams& p, float x_tan, float& sl,
                                               float& sr)
{
    const float r = p.r, rr = p.rr, fmh = p.fmh, tr = p.tr, tl = p.tl;
    const float fx = p.f * x_tan;
    float xr = p.w - ((fx - p.w) * p.h) / fmh;
    float xm = ((-fx) * p.h) / fmh;
    float xl = (-p.w) - ((fx + p.w) * p.h) / fmh;
    xr = clampf(xr, -0.5f, 0.5f); xm = clampf(xm, -0.5f, 0.5f); xl = clampf(xl, -0.5f, 0.5f);
    float ur = __ocml_acos_f32(xr / r), um = __ocml_acos_f32(xm / r), ul = __ocml_acos_f32(xl / r);
    float sm = seg(um);
    float sr_ml = rr * (sm - seg(ur));
    float sl_ml = rr * (seg(ul) - sm);
    float ure = clampf(ur, tr, tl), ume = clampf(um, tr, tl), ule = clampf(ul, tr, tl);
    float xre = __ocml_cos_f32(ure) * r, xme = __ocml_cos_f32(ume) * r, xle = __ocml_cos_f32(ule) * r;
    float sme = seg(ume);
    sr_ml = sr_ml - ((rr * (sme - seg(ure))) - (xre - xme));
    sl_ml = sl_ml - ((rr * (seg(ule) - sme)) - (xme - xle));

    const float hx = p.h * x_tan;
    xr = p.w - hx; xm = 0.0f - hx; xl = (-p.w) - hx;
    xr = clampf(xr, -0.5f, 0.5f); xm = clampf(xm, -0.5f, 0.5f); xl = clampf(xl, -0.5f, 0.5f);
    ur = __ocml_acos_f32(xr / r); um = __ocml_acos_f32(xm / r); ul = __ocml_acos_f32(xl / r);
    sm = seg(um);
    float sr_in = rr * (sm - seg(ur));
    float sl_in = rr * (seg(ul) - sm);
    ure = clampf(ur, tr, tl); ume = clampf(um, tr, tl); ule = clampf(ul, tr, tl);
    xre = __ocml_cos_f32(ure) * r; xme = __ocml_cos_f32(ume) * r; xle = __ocml_cos_f32(ule) * r;
    sme = seg(ume);
    sr_in = sr_in - ((rr * (sme - seg(ure))) - (xre - xme));
    sl_in = sl_in - ((rr * (seg(ule) - sme)) - (xme - xle));
    const float sr_mg = (xr - xm) * 1.0f - sr_in;
    const float sl_mg = (xm - xl) * 1.0f - sl_in;
    sr = sr_ml + sr_mg;
    sl = sl_ml + sl_mg;
}

// Splat geometry of one sensor-plane ray (monte_carlo.py:24-38, 209-222):
// window test, bilinear taps.  Returns false when the ray carries no weight.
struct SplatTaps {
    int i_tl, i_tr, i_bl, i_br;     // linear indices into a ks*ks tile
    float w_tl, w_tr, w_bl, w_br;   // bilinear weight * ra
};

struct SplatGeom {
    float lim;      // fp32((ks/2-0.5)*ps - 0.01*ps)              monte_carlo.py:37
    float x_min;    // fp32((-ks/2+0.5)*ps)
    float y_max;    // fp32(( ks/2-0.5)*ps)
    float dx_rng;   // fp32(x_max - x_min)
    float dy_rng;   // fp32(y_min - y_max)
    float ksm1;     // ks - 1
    int32_t ks;
};

__device__ __forceinline__ bool splat_taps(const SplatGeom& gm, float sx, float sy, float cx,
                                           float cy, float ra, SplatTaps& tp)
{
    float px = (-sx) - cx;                     // points = -o.xy ; points - pointc_ref
    float py = (-sy) - cy;
    float w = ra * (__builtin_fabsf(px) < gm.lim ? 1.0f : 0.0f);
    w = w * (__builtin_fabsf(py) < gm.lim ? 1.0f : 0.0f);
    if (!(w != 0.0f)) return false;            // adds exact zeros in the reference
    px = px * w; py = py * w;                  // :38
    const float pf0 = ((py - gm.y_max) / gm.dy_rng) * gm.ksm1;   // row
    const float pf1 = ((px - gm.x_min) / gm.dx_rng) * gm.ksm1;   // col
    const float fl0 = __builtin_floorf(pf0), fl1 = __builtin_floorf(pf1);
    const float wb = pf0 - fl0, wr = pf1 - fl1;
    const int r0 = (int)fl0, c0 = (int)fl1;
    const int r1 = (int)__builtin_floorf(pf0 + 1.0f), c1 = (int)__builtin_floorf(pf1 + 1.0f);
    const int ks = gm.ks;
    tp.i_tl = r0 * ks + c0;       tp.w_tl = ((1.0f - wb) * (1.0f - wr)) * w;
    tp.i_tr = r0 * ks + c1;       tp.w_tr = ((1.0f - wb) * wr) * w;
    tp.i_bl = r1 * ks + c0;       tp.w_bl = (wb * (1.0f - wr)) * w;
    tp.i_br = (r0 + 1) * ks + (c0 + 1); tp.w_br = (wb * wr) * w;
    // The window test keeps every tap inside [0, ks-1]; guard anyway so that a
    // NaN position can never write outside the tile.
    const bool ok = r0 >= 0 && c0 >= 0 && r1 < ks && c1 < ks && (r0 + 1) < ks && (c0 + 1) < ks;
    return ok;
}

}  // namespace sdirt
