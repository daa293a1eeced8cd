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

#include <type_traits>

namespace sdirt {

constexpr int kMaxAi = 8;
constexpr int kPeriodicFrom = 4;        // Newton tables longer than this run the periodicity test
constexpr float kNewtonStepBound = 5.0f;   // deeplens/surfaces.py:29

// ---------------------------------------------------------------------------
// Per-surface constants in device memory: two 64-byte blocks per surface, laid out for the
// scalar unit.  A wave reads the 12 dwords of block `h` it needs with one scalar round trip per
// surface (Surf / surf_issue below: into SGPRs, no VGPRs, no LDS bandwidth) and block `p` with one
// more on aspheres only.  Every field the
// reference obtains by rounding a Python/numpy float64 to fp32 at the point of use is rounded on
// the host, once, in sdirt_lens_create (sdirt_dp.hip: make_dev_surface).
// ---------------------------------------------------------------------------
struct alignas(64) SurfHot {
    uint32_t flags;      // bits 0-1 kind (0 plane, 1 sphere, 2 asphere) | 2 do_refract (plane: eta != 1,
                         // surfaces.py:450) | 3 k > -1 (:727,738) | 4 c > 0 | 5 1 + k == 1 | 8-11 ai_degree
    float d, c;
    float c2;            // c*c (fp32)
    float onepk;         // 1 + k
    float lim_loose;     // ((1/c2) * fp32(1-1e-9)) / (1+k)   surfaces.py:728,739
    float r2_lim;        // fp32(r*r in double)               surfaces.py:464,728
    float lim_tight;     // curved: min(r2_lim, lim_loose) when k > -1, else r2_lim -- the two domain
                         // tests of _valid (surfaces.py:727-733) as one; planes: fp32(r) (surfaces.py:421)
    // dwords 8-11: what refraction needs when tracing forward; 12-15: the same for backward
    float d_plus_R;      // d + 1/c                           surfaces.py:607-615
    float eta_f, eta2_f; // forward: fp32(n1/n2), fp32((n1/n2)^2)   surfaces.py:401,663-669
    float k;
    float d_plus_R_b;    // = d_plus_R
    float eta_b, eta2_b; // backward: fp32(n2/n1), fp32((n2/n1)^2)  surfaces.py:404
    uint32_t pad;
};
struct alignas(64) SurfPoly {
    float ai[kMaxAi];    // ai2, ai4, ...
    float kai[kMaxAi];   // (i+1) * ai[i]  (python int * fp32 tensor)  surfaces.py:823
};
struct DevSurface {
    SurfHot h;
    SurfPoly p;
};
static_assert(sizeof(SurfHot) == 64 && sizeof(SurfPoly) == 64 && sizeof(DevSurface) == 128, "layout");

constexpr uint32_t kFlagRefract = 4u, kFlagKgtM1 = 8u, kFlagCpos = 16u;
constexpr uint32_t kFlagUnitK = 32u;      // 1 + k == 1.0f exactly (every sphere): (1 + k) * r2 is r2

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// One 64-byte block through the scalar cache into 16 consecutive SGPRs, waited for on the spot.
// The address is wave-uniform by construction.
__device__ __forceinline__ u32x16 sload_block(const void* p)
{
    u32x16 r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
    return r;
}

// The constants of one surface in registers: 12 SGPRs, fetched with ONE scalar-memory round trip
// per surface (two loads, one wait).  Left to the compiler, the fields of a `const DevSurface&`
// are fetched where they are first used: seven dependent s_load / s_waitcnt round trips (and one
// vector load for the trip count) per surface in round 1's ISA, a quarter of every wave's time.
struct Surf {
    u32x8 a;      // flags d c c2 onepk lim_loose r2_lim lim_tight
    u32x4 b;      // d_plus_R eta eta2 (direction-specific)
    __device__ __forceinline__ int kind() const { return (int)(a[0] & 3u); }
    __device__ __forceinline__ int ai_degree() const { return (int)((a[0] >> 8) & 15u); }
    __device__ __forceinline__ bool do_refract() const { return (a[0] & kFlagRefract) != 0u; }
    __device__ __forceinline__ bool k_gt_m1() const { return (a[0] & kFlagKgtM1) != 0u; }
    __device__ __forceinline__ bool c_pos() const { return (a[0] & kFlagCpos) != 0u; }
    __device__ __forceinline__ bool unit_k() const { return (a[0] & kFlagUnitK) != 0u; }
    __device__ __forceinline__ float d() const { return __uint_as_float(a[1]); }
    __device__ __forceinline__ float c() const { return __uint_as_float(a[2]); }
    __device__ __forceinline__ float c2() const { return __uint_as_float(a[3]); }
    __device__ __forceinline__ float onepk() const { return __uint_as_float(a[4]); }
    __device__ __forceinline__ float lim_loose() const { return __uint_as_float(a[5]); }
    __device__ __forceinline__ float r2_lim() const { return __uint_as_float(a[6]); }
    __device__ __forceinline__ float lim_tight() const { return __uint_as_float(a[7]); }
    __device__ __forceinline__ float r_lim() const { return __uint_as_float(a[7]); }     // planes
    __device__ __forceinline__ float d_plus_R() const { return __uint_as_float(b[0]); }
    __device__ __forceinline__ float eta() const { return __uint_as_float(b[1]); }
    __device__ __forceinline__ float eta2() const { return __uint_as_float(b[2]); }
    __device__ __forceinline__ struct Poly poly(const DevSurface* blk) const;   // the asphere's polynomial block
};

// One surface's constants in flight: the two loads of its block plus the dword of the launch's
// Newton trip table that holds its count (trip_words: the table in the kernel-argument segment,
// one signed byte per surface).  surf_issue() starts the three scalar loads, surf_wait() is the
// s_waitcnt that makes the registers readable -- the trace loop issues surface k+1's loads in the
// middle of surface k (after its Newton solve, before its refraction) and waits at the end of
// surface k, so the scalar-cache latency (which the wave otherwise sits out once per surface:
// measured as a 3.5 % loss on the 21-surface rf35mm) is covered by ~100 vector instructions.
// Between issue and wait the registers must not be touched: they are outputs of one asm statement
// and in/outs of the next, with no other use in between.
struct SurfRaw {
    u32x8 a;
    u32x4 b;
    uint32_t tw;
};

template <bool FWD>
__device__ __forceinline__ void surf_issue(SurfRaw& n, const DevSurface* blk, const void* trip_words, int k)
{
    const uint32_t off = (uint32_t)(k >> 2) << 2;
    if (FWD)
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dword %2, %4, %5"
                     : "=&s"(n.a), "=&s"(n.b), "=&s"(n.tw) : "s"(blk), "s"(trip_words), "s"(off) : "memory");
    else
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x30\n\ts_load_dword %2, %4, %5"
                     : "=&s"(n.a), "=&s"(n.b), "=&s"(n.tw) : "s"(blk), "s"(trip_words), "s"(off) : "memory");
}

__device__ __forceinline__ void surf_wait(SurfRaw& n)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n.a), "+s"(n.b), "+s"(n.tw) : : "memory");
}

__device__ __forceinline__ int surf_trips(const SurfRaw& n, int k)
{
    return (int)(int8_t)(n.tw >> ((k & 3) * 8));
}

// The polynomial block in registers (aspheres only): ai(i) = w[i], kai(i) = w[8 + i].
struct Poly {
    u32x16 w;
    __device__ __forceinline__ float ai(int i) const { return __uint_as_float(w[i]); }
    __device__ __forceinline__ float kai(int i) const { return __uint_as_float(w[8 + i]); }
};
struct NoPoly {};
__device__ __forceinline__ Poly Surf::poly(const DevSurface* blk) const
{
    Poly p;
    p.w = sload_block(&blk->p);
    return p;
}

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

struct Ray {
    float ox, oy, oz, dx, dy, dz, ra, ob;
};

// Math policies.
//  Ieee: correctly rounded / and sqrt (hipcc's default expansion: 12- and 17-instruction
//        sequences with range scaling and special-case fix-up): bit-identical to an IEEE CPU
//        evaluation of the reference's operation sequence for ALL operands.
//  Lean: the same correctly rounded results for every NORMAL-RANGE operand, without the
//        range scaling (v_div_scale / 2^32 pre-scaling) and special-value fix-up
//        (v_div_fixup, class tests) that the compiler's sequences carry for denormal, huge
//        and zero/inf operands:
//          div : y0 = rcp(b); y = y0 + y0*(1 - b*y0); q0 = a*y; q = q0 + y*(a - b*q0)
//                (all via fma; 5 + v_rcp).  Verified bit-identical to IEEE on ALL 2^46 mantissa
//                pairs (sdirt_selftest_math mode 2; the recurrence is exact-scaling in the
//                exponent, so this covers every operand pair whose exponents stay within +-60).
//                a == 0 gives the correctly signed zero; b == 0 gives NaN instead of inf.
//          sqrt: s = v_sqrt_f32(x) (<= 1 ulp), then pick among {s-ulp, s, s+ulp} by the sign
//                of the exact residuals x - s'*s (fma) -- the compiler's own correction
//                step (10 + v_sqrt).  Verified on EVERY positive normal fp32 (mode 0,
//                exhaustive); +-0, +inf, negative and NaN inputs behave as IEEE; denormal
//                inputs do not.
//          sqrt_pos: for arguments KNOWN to be positive, finite and normal (1 - a inside the
//                conic's domain, squared lengths of non-zero vectors): r = v_rsq_f32(x);
//                s = x*r; s += (r/2)*(x - s*s) with the residual formed exactly by one fma
//                (4 + v_rsq).  The Goldschmidt refinements of s and r/2 that LLVM's own
//                flush-denormal sqrt expansion carries in between (3 more fma) change no result:
//                verified on every fp32 in [2^-100, 2^100] with and without them (mode 3,
//                exhaustive); 0, inf and denormals are NOT handled.  (The analogous shortening of
//                the division -- unrefined v_rcp seed, one residual correction -- is NOT exact:
//                47045 of the 2^46 mantissa pairs miss the IEEE quotient.)
//        No operand on a valid ray is denormal or zero-denominator (eps = 1e-9 guards, unit
//        direction vectors, |positions| in [1e-6, 2e4] mm), so valid rays are bit-identical
//        to the Ieee instantiation; tests/test_gpu_parity.py runs both against the oracle.
// Exact fusions (Lean only; Ieee keeps the reference's literal operation sequence).  A product with a power
// of two or with a 0/1 flag is EXACT (no rounding, barring over/underflow), so the reference's
// "round the product, then add" is what one fma computes:
//   dfdt(dgd, y, dz)      : dgd * (2*y) - dz          = fma(2, dgd*y, -dz)        [2*(p) commutes with rounding]
//   add_half_quot(s, a, b): s + (a*0.5)/b             = fma(0.5, a/b, s)          [(a/2)/b == (a/b)/2]
//   one_minus_flagged(x,f): 1 - x*f,  f in {0, 1}     = fma(-x, f, 1)
// One vector instruction less each, 2 per sag evaluation (49 evaluations per ray) and 1 per refraction.
// Outside the normal range (|values| below 2^-125: a denormal product) the two forms can round differently;
// no traced ray gets there (positions are sums of O(1)-mm terms: 0 or >= 1e-7 mm), and the parity tests
// run both policies against the oracle.
//   newton_step(ft, b)    : clamp(ft / b, -5, 5) (surfaces.py:557-559).  torch.clamp passes a NaN through; one
//        v_med3_f32 returns a bound for it.  With finite rays the quotient is NaN only when b = dfdt + 1e-9 is
//        exactly 0 -- where the reference divides to +-inf and steps by +-5, and Lean::div (which trades the
//        inf for a NaN, see below) is not the reference anyway.  A ray that ENTERS a surface with a NaN
//        position keeps a NaN t whatever the step is (t0 is NaN).  So under Lean the NaN is not put back:
//        two vector instructions (v_cmp_u, v_cndmask) less per sag evaluation; Ieee puts it back.
//   keep_if(ra, v)         : ra * (v ? 1 : 0) = v ? ra : 0 for every finite weight.
struct Ieee {
    static constexpr bool kFused = false;
    static __device__ __forceinline__ float newton_step(float ft, float b)
    {
        const float q = ft / b;
        const float m = __builtin_amdgcn_fmed3f(q, -kNewtonStepBound, kNewtonStepBound);
        return q != q ? q : m;
    }
    static __device__ __forceinline__ float keep_if(float ra, bool v) { return ra * (v ? 1.0f : 0.0f); }
    static __device__ __forceinline__ float dfdt(float dgd, float y, float dz) { return dgd * (2.0f * y) - dz; }
    static __device__ __forceinline__ float add_half_quot(float s, float a, float b) { return s + (a * 0.5f) / b; }
    static __device__ __forceinline__ float one_minus_flagged(float x, float f) { return 1.0f - x * f; }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
    static __device__ __forceinline__ float sqrt_pos(float x) { return __builtin_sqrtf(x); }
    static __device__ __forceinline__ void div3(float& a0, float& a1, float& a2, float b)
    {
        a0 = a0 / b; a1 = a1 / b; a2 = a2 / b;
    }
};
struct Lean {
    static constexpr bool kFused = true;
    static __device__ __forceinline__ float newton_step(float ft, float b)
    {
        return __builtin_amdgcn_fmed3f(div(ft, b), -kNewtonStepBound, kNewtonStepBound);
    }
    static __device__ __forceinline__ float keep_if(float ra, bool v) { return v ? ra : 0.0f; }
    static __device__ __forceinline__ float dfdt(float dgd, float y, float dz)
    {
        return __builtin_fmaf(2.0f, dgd * y, -dz);
    }
    static __device__ __forceinline__ float add_half_quot(float s, float a, float b)
    {
        return __builtin_fmaf(0.5f, div(a, b), s);
    }
    static __device__ __forceinline__ float one_minus_flagged(float x, float f) { return __builtin_fmaf(-x, f, 1.0f); }
    static __device__ __forceinline__ float div(float a, float b)
    {
        const float y0 = __builtin_amdgcn_rcpf(b);
        const float y = __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
        const float q0 = a * y;
        return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    }
    // three numerators over one denominator: the refined reciprocal is computed once; each
    // quotient is the same arithmetic as div(), hence the same bits
    static __device__ __forceinline__ void div3(float& a0, float& a1, float& a2, float b)
    {
        const float y0 = __builtin_amdgcn_rcpf(b);
        const float y = __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
        float q = a0 * y; a0 = __builtin_fmaf(__builtin_fmaf(-b, q, a0), y, q);
        q = a1 * y;       a1 = __builtin_fmaf(__builtin_fmaf(-b, q, a1), y, q);
        q = a2 * y;       a2 = __builtin_fmaf(__builtin_fmaf(-b, q, a2), y, q);
    }
    static __device__ __forceinline__ float sqrt(float x)
    {
        float s = __builtin_amdgcn_sqrtf(x);
        const float sm = __uint_as_float(__float_as_uint(s) - 1u);
        const float sp = __uint_as_float(__float_as_uint(s) + 1u);
        const float rm = __builtin_fmaf(-sm, s, x);          // x - (s - ulp) * s
        const float rp = __builtin_fmaf(-sp, s, x);          // x - (s + ulp) * s
        s = rm <= 0.0f ? sm : s;
        s = rp > 0.0f ? sp : s;
        // +-0 and +inf map to themselves; negative / NaN inputs give NaN through v_sqrt_f32
        return __builtin_amdgcn_classf(x, 0x260) ? x : s;   // -0 | +0 | +inf
    }
    static __device__ __forceinline__ float sqrt_pos(float x)
    {
        const float r = __builtin_amdgcn_rsqf(x);
        const float s = x * r;
        const float h = 0.5f * r;
        const float d = __builtin_fmaf(-s, s, x);            // exact residual of the 1-ulp estimate
        return __builtin_fmaf(d, h, s);
    }
    // div() in two halves: div(a, b) == div_y(a, b, recip(b)).  (The seed must be v_rcp_f32(b):
    // seeding the Newton step from a related quantity instead -- the square root's own
    // half-reciprocal for a / sqrt(x), the square of 1/b's reciprocal for a / (b*b) -- was tried to
    // save the v_rcp_f32 and gives a reciprocal that differs in the last bit for 80 + 6 arguments,
    // of which the all-ones-mantissa divisors then miss the IEEE quotient for some numerators.)
    static __device__ __forceinline__ float recip(float b)
    {
        const float y0 = __builtin_amdgcn_rcpf(b);
        return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
    }
    static __device__ __forceinline__ float div_y(float a, float b, float y)
    {
        const float q0 = a * y;
        return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    }
};

// a / b for a divisor that is the same for every ray of the launch: the refined reciprocal y of
// Lean::div is computed once (UDiv::make), each quotient is Lean::div's remaining three
// operations -- the same arithmetic, hence the same (correctly rounded) bits.  Ieee: plain a / b.
template <class M> struct UDiv;
template <> struct UDiv<Ieee> {
    float b;
    static __device__ __forceinline__ UDiv make(float b) { return UDiv{b}; }
    __device__ __forceinline__ float operator()(float a) const { return a / b; }
};
template <> struct UDiv<Lean> {
    float b, y;
    static __device__ __forceinline__ UDiv make(float b) { return UDiv{b, Lean::recip(b)}; }
    __device__ __forceinline__ float operator()(float a) const { return Lean::div_y(a, b, y); }
};

__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    // torch.clamp semantics: NaN propagates
    if (v != v) return v;
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

// torch.clamp(v, lo, hi) for a value known not to be NaN (one v_med3_f32): the sub-pixel
// boundaries of a LIVE ray -- dead rays never reach the weights (splat_taps)
__device__ __forceinline__ float clamp_finite(float v, float lo, float hi)
{
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}

// torch.nn.functional.normalize over a last dim of 3 (basics.py:245,
// surfaces.py:628): v / max(||v||, 1e-12); torch's CPU kernel accumulates the
// squares with fused multiply-adds (x*x, then fma y, then fma z).
// NONZERO: the vector is known not to vanish (a direction between distinct points, a surface
// normal with a -1 / 2(z - centre) component): sqrt_pos applies and the 1e-12 floor is dead.
template <class M, bool NONZERO = false>
__device__ __forceinline__ void normalize3(float& x, float& y, float& z)
{
    float acc = x * x;
    acc = __builtin_fmaf(y, y, acc);
    acc = __builtin_fmaf(z, z, acc);
    float nrm = NONZERO ? M::sqrt_pos(acc) : M::sqrt(acc);
    if (!NONZERO) nrm = nrm < 1e-12f ? 1e-12f : nrm;      // a non-zero vector's norm is >= 1e-19
    M::div3(x, y, z, nrm);
}

// Even-asphere sag g(r2) and dg/d(r2), surfaces.py:787-808 and 811-830 evaluated together
// (they share sqrt(1-a)).  INSIDE: r2 is known to lie inside the conic's domain with k > -1
// (0 < 1 - a <= 1), so sqrt_pos applies.  P = Poly: the surface has polynomial terms (deg of
// them, wave-uniform); P = NoPoly: pure conic (every sphere).
// r2 ** n as torch evaluates it on CPU: n == 2 -> x*x, n == 3 -> (x*x)*x, n >= 4 a <= 1-ulp
// vector pow, for which the correctly rounded exact power stands here (running product in
// fp64, rounded once per term) -- the parity oracle evaluates the same expression.
// the conic's constants as the Newton loop holds them: in VGPRs.  A vector instruction with an
// SGPR source operand issues at 0.55x the rate of one with VGPR / inline-constant sources on
// gfx950 (tools/form_bench.hip: 4.2 vs 2.25 cycles per SIMD at 8 waves); six per trip against
// four v_mov per surface.  (End to end the two forms time the same within run-to-run noise:
// profiles/r02 kbench logs, profiles/r03/k_psf_lr_sites.txt; tools/variants/form_conic_sgpr.py builds the SGPR form.)
struct ConicV {
    float c, c2, onepk, d;
};
__device__ __forceinline__ float to_vgpr(float s)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
template <class S>
__device__ __forceinline__ ConicV conic_v(const S& s)
{
    return ConicV{to_vgpr(s.c()), to_vgpr(s.c2()), to_vgpr(s.onepk()), to_vgpr(s.d())};
}
struct ConicS {      // the same constants straight from SGPRs (cold paths)
    float c, c2, onepk, d;
};
template <class S>
__device__ __forceinline__ ConicS conic_s(const S& s) { return ConicS{s.c(), s.c2(), s.onepk(), s.d()}; }

template <class M, bool INSIDE, class C, bool UNITK = false>
__device__ __forceinline__ void sag_g_dgd(const C& k, const NoPoly&, int, float r2, float& g, float& dgd)
{
    // UNITK: 1 + k is 1.0f, the product with it is the identity and is left out
    const float a = UNITK ? r2 * k.c2 : (k.onepk * r2) * k.c2;
    const float sf = INSIDE ? M::sqrt_pos(1.0f - a) : M::sqrt(1.0f - a);
    const float onesf = 1.0f + sf;
    g = M::div(r2 * k.c, onesf);
    dgd = M::div(M::add_half_quot(onesf, a, sf) * k.c, onesf * onesf);   // (onesf + (a/2)/sf) * c / onesf^2
}

template <class M, bool INSIDE, class C, bool UNITK = false, class P = Poly>
__device__ __forceinline__ void sag_g_dgd(const C& k, const P& pol, int deg, float r2, float& g, float& dgd)
{
    sag_g_dgd<M, INSIDE, C, UNITK>(k, NoPoly{}, 0, r2, g, dgd);
    dgd = dgd + pol.ai(0);
    g = g + pol.ai(0) * r2;
    float pw = r2;                        // r2 ** i
    const double xd = (double)r2;
    double accd = xd;                     // r2 ** n in fp64: ((x*x)*x)*...
#pragma unroll
    for (int i = 1; i < kMaxAi; ++i) {
        if (i < deg) {
            dgd = dgd + pol.kai(i) * pw;
            const int n = i + 1;
            if (deg > 3) accd = accd * xd;
            pw = n == 2 ? r2 * r2 : n == 3 ? (r2 * r2) * r2 : (float)accd;
            g = g + pol.ai(i) * pw;
        }
    }
}

// surfaces.py:523-586.  `trips` loop iterations (wave-uniform), then the extra
// differentiable step and the validity test.  Returns Newton's own validity;
// mask_out (wave-uniform, lives in SGPRs) gets bit j set when ANY active lane of
// the wave had |f(t)| > 50e-6 in trip j -- the per-wave share of the reference's
// batch-wide `.any()` loop condition (surfaces.py:547).
template <class M, bool KGT, class P, bool UNITK = false, class S = Surf>
__device__ __forceinline__ bool newton_k(const S& s, const P& pol, const Ray& r, int trips, float& t_out,
                                         uint32_t& mask_out)
{
    using CV = ConicV;
    const bool adaptive = trips < 0;
    const int cap = adaptive ? -trips : trips;
    const float tol_loose = (float)50e-6, tol_tight = (float)10e-6, eps = (float)1e-9;
    const ConicV k = conic_v(s);
    const int deg = s.ai_degree();
    const float t0 = M::div(k.d - r.oz, r.dz);
    const float dd = r.dx * r.dx + r.dy * r.dy;
    const float dox = r.dx * r.ox + r.dy * r.oy;
    const bool alive = r.ra > 0.0f;
    // `valid_loose && ra > 0` as ONE comparison per trip: a dead ray compares against a bound no
    // squared radius can pass (k > -1: rr < lim_loose; k <= -1: rr > 0, surfaces.py:736-743)
    const float bound = KGT ? (alive ? s.lim_loose() : -1.0f) : (alive ? 0.0f : __builtin_inff());
    float t = t0;
    // Loop bookkeeping in four scalar instructions per trip (the compiler's own rendering of
    // `mask |= open ? bit : 0; bit <<= 1; if (adaptive && !open) break;` took thirteen and three
    // branches): SCC = "some ray of this wave is open"; it is shifted into `rmask` from the right
    // (s_addc: rmask + rmask + SCC), so the trips sit in REVERSE order there, and in adaptive mode
    // a closed wave clears the count of trips left.
    // trips >= 0: exactly that many trips (the reference's batch-wide count, supplied by the host).
    // trips < 0: at most -trips, and this WAVE stops as soon as none of its 64 rays is open -- the
    // reference's loop condition evaluated per wave instead of per batch (speed mode, no host check).
    uint32_t rmask = 0;
    const uint32_t amask = adaptive ? ~0u : 0u;
    int left = cap;
    // Exact early exit.  A trip maps t to F(t) and nothing else changes, so once t[n+1] == t[n]
    // (a fixed point) or t[n+1] == t[n-1] (the two-value flip of a last bit), bit for bit, the
    // sequence is periodic and the remaining trips can be written down instead of run: with `r`
    // of them left the final t is t[n+1] (r even) or t[n] (r odd), and the wave's open flags
    // repeat with period two.
    // When EVERY ray of the wave is there, the wave leaves the loop.  (On the first surface the
    // reference runs to its 10-trip cap for distant objects -- t ~ 2e4 mm resolves the surface to
    // 1e-3 mm only, |f| never gets below 50e-6 -- while every wave is periodic after 2 to 4.)
    uint32_t tp = 0xffffffffu;            // t two trips back: none yet (no trip produces this NaN)
    float t_odd = t;
    int skipped = 0;
    unsigned long long open_cur = 0ull, open_prev = 0ull, fixed = 0ull;   // lane masks of the last trips
    auto trip = [&](auto periodic_exit) __attribute__((always_inline)) {
        left -= 1;
        const float nx = r.ox + r.dx * t, ny = r.oy + r.dy * t, nz = r.oz + r.dz * t;
        const float rr = nx * nx + ny * ny;
        // g(x*valid, y*valid) (surfaces.py:694-696): (x*1)^2 + (y*1)^2 is rr bit for bit, and
        // (x*0)^2 + (y*0)^2 is 0 for every finite position
        const float r2 = (KGT ? rr < bound : rr > bound) ? rr : 0.0f;
        float g, dgd;
        sag_g_dgd<M, KGT, CV, UNITK>(k, pol, deg, r2, g, dgd);
        const float ft = (g + k.d) - nz;
        const float dfdt = M::dfdt(dgd, dd * t + dox, r.dz);            // dgd * dr2dt - dz, dr2dt = 2((dx^2+dy^2) t + (dx ox + dy oy))
        const unsigned long long open = __ballot(__builtin_fabsf(ft) > tol_loose);
        const float tn = t - M::newton_step(ft, dfdt + eps);
        uint32_t tmp;
        if (decltype(periodic_exit)::value) {
            open_prev = open_cur;
            open_cur = open;
            asm volatile("v_cmp_eq_u32 vcc, %7, %8\n\t"          // t[n+1] == t[n-1]: period two
                         "v_cmp_eq_u32_e64 %4, %7, %9\n\t"       // t[n+1] == t[n]:   fixed point
                         "s_cmp_lg_u64 %5, 0\n\t"                // SCC = open
                         "s_cselect_b32 %3, 0, %6\n\t"
                         "s_addc_u32 %0, %0, %0\n\t"
                         "s_andn2_b32 %1, %1, %3\n\t"
                         "s_or_b64 vcc, vcc, %4\n\t"
                         "s_cmp_eq_u64 vcc, exec\n\t"            // SCC = every ray of the wave is periodic
                         "s_cselect_b32 %2, %1, %2\n\t"
                         "s_cselect_b32 %1, 0, %1"
                         : "+s"(rmask), "+s"(left), "+s"(skipped), "=&s"(tmp), "=&s"(fixed)
                         : "s"(open), "s"(amask), "v"(tn), "v"(tp), "v"(t) : "scc", "vcc");
            tp = __float_as_uint(t);
            t_odd = t;
        } else {
            asm volatile("s_cmp_lg_u64 %3, 0\n\t"
                         "s_cselect_b32 %2, 0, %4\n\t"
                         "s_addc_u32 %0, %0, %0\n\t"
                         "s_andn2_b32 %1, %1, %2"
                         : "+s"(rmask), "+s"(left), "=&s"(tmp) : "s"(open), "s"(amask) : "scc");
        }
        t = tn;
    };
    // the periodicity test costs four vector and five scalar instructions per trip: it is run
    // where it can pay, on tables longer than kPeriodicFrom trips (a wave-uniform choice of loop)
    if (cap > kPeriodicFrom) {
        while (left > 0) trip(std::true_type{});
    } else
    {
        while (left > 0) trip(std::false_type{});
    }
    if (skipped > 0) {
        // the open flags of the skipped trips alternate A, B, A, ...: B = the last flag (the state
        // t[n]); A belongs to the state t[n+1], which a fixed-point ray shares with t[n] and a
        // period-two ray with t[n-1]
        const bool a_open = ((open_cur & fixed) | (open_prev & ~fixed)) != 0ull;
        const uint32_t pat = (a_open ? 0xAAAAu : 0u) | ((rmask & 1u) ? 0x5555u : 0u);
        rmask = (rmask << skipped) | (pat >> (16 - skipped));
        t = (skipped & 1) ? t_odd : t;
    }
    // trip j (0-based) -> bit j + 1.  n trips ran: cap, or in adaptive mode up to and including
    // the first closed one (rmask = 1..10 then; an all-open run ends in a 1)
    uint32_t n = (uint32_t)cap;
    if (adaptive && !(rmask & 1u)) n = 32u - (uint32_t)__builtin_clz(rmask | 1u);
    const uint32_t mask = __builtin_bitreverse32(rmask) >> (31u - n);
    mask_out = mask;
    const float t1 = t - t0;   // :563
    t = t0 + t1;               // :567
    float nx = r.ox + r.dx * t, ny = r.oy + r.dy * t;
    const float nz = r.oz + r.dz * t;
    float rr = nx * nx + ny * ny;
    // _valid (surfaces.py:724-733): rr < r2_lim [and rr < lim_loose] = rr < lim_tight
    const float tight = alive ? s.lim_tight() : -1.0f;
    const float r2 = rr < tight ? rr : 0.0f;
    float g, dgd;
    sag_g_dgd<M, KGT, CV, UNITK>(k, pol, deg, r2, g, dgd);
    const float ft = (g + k.d) - nz;
    const float dfdt = M::dfdt(dgd, dd * t + dox, r.dz);
    t = t - M::newton_step(ft, dfdt + eps);
    nx = r.ox + r.dx * t;
    ny = r.oy + r.dy * t;
    rr = nx * nx + ny * ny;
    bool v = rr < tight;
    v = v && __builtin_fabsf(ft) < tol_tight;
    v = v && t > 0.0f;
    t_out = t;
    return v;
}

// surfaces.py:633-679 with _normal (:589-630).  FWD: rays travel +z (n negated,
// eta = n1/n2); !FWD: backward tracing.
template <bool FWD, class M, class P, class S>
__device__ __forceinline__ void refract(const S& s, const P& pol, Ray& r)
{
    float nx, ny, nz;
    float sgn = 1.0f;              // sign of the normal still to be applied (spheres under the Lean policy)
    bool flip_fwd = FWD;
    if (!std::is_same<P, NoPoly>::value) {
        const float vf = r.ra > 0.0f ? 1.0f : 0.0f;
        const float xv = r.ox * vf, yv = r.oy * vf;
        float g, ds;
        // _dsdr2 is evaluated at whatever (x, y) the ray holds (surfaces.py:600): general sqrt
        sag_g_dgd<M, false>(conic_s(s), pol, s.ai_degree(), xv * xv + yv * yv, g, ds);
        nx = (ds * 2.0f) * xv; ny = (ds * 2.0f) * yv; nz = -1.0f;
    } else if (s.kind() == 0) {
        nx = 0.0f; ny = 0.0f; nz = -1.0f;
    } else if (s.kind() == 1 && M::kFused) {
        // sphere: n = normalize(+-2 (o - centre)), negated when tracing forward (surfaces.py:607-615, 650).
        // Scaling by +-2 is exact and commutes with every rounding of the normalisation, so
        // n = sgn * m with m = normalize(o - centre) -- and the sign only survives in the sr * n term
        // below (cosi * n = (sgn cm)(sgn m) = cm m): three multiplications by +-2 and the negation
        // become one multiplication of sr by +-1.
        nx = r.ox; ny = r.oy; nz = r.oz - s.d_plus_R();
        sgn = (s.c_pos() != FWD) ? 1.0f : -1.0f;
        flip_fwd = false;
    } else if (s.kind() == 1) {
        const float sg = s.c_pos() ? 2.0f : -2.0f;       // (+-2) * x is exact either way
        nx = sg * r.ox; ny = sg * r.oy; nz = sg * r.oz - sg * s.d_plus_R();
    } else {                                              // conic without polynomial terms
        const float vf = r.ra > 0.0f ? 1.0f : 0.0f;
        const float xv = r.ox * vf, yv = r.oy * vf;
        float g, ds;
        sag_g_dgd<M, false>(conic_s(s), pol, 0, xv * xv + yv * yv, g, ds);
        nx = (ds * 2.0f) * xv; ny = (ds * 2.0f) * yv; nz = -1.0f;
    }
    // a surface normal never vanishes on a ray that sits on the surface; the radicand of `sr` is
    // >= 2^-24 by the validity test (sqrt_pos; a dead ray's garbage stays confined to the
    // discarded candidate direction)
    normalize3<M, true>(nx, ny, nz);
    if (flip_fwd) { nx = -nx; ny = -ny; nz = -nz; }
    const float eta = s.eta(), eta2 = s.eta2();
    const float cosi = (r.dx * nx + r.dy * ny) + r.dz * nz;     // up to the pending sign
    const float c2i = cosi * cosi;
    const float omc = 1.0f - c2i;
    const bool v = c2i > 0.1f && eta2 * omc < 1.0f && r.ra > 0.0f;
    const float vf = v ? 1.0f : 0.0f;
    float sr = M::sqrt_pos(M::one_minus_flagged(eta2 * omc, vf));
    if (M::kFused && s.kind() == 1) sr = sr * sgn;
    float ndx = sr * nx + eta * (r.dx - cosi * nx);
    float ndy = sr * ny + eta * (r.dy - cosi * ny);
    float ndz = sr * nz + eta * (r.dz - cosi * nz);
    ndx = v ? ndx : r.dx; ndy = v ? ndy : r.dy; ndz = v ? ndz : r.dz;
    r.ob = r.ob * ((ndx * r.dx + ndy * r.dy) + ndz * r.dz);
    r.dx = ndx; r.dy = ndy; r.dz = ndz;
    r.ra = r.ra * vf;
}

// Curved surface: Newton intersection, validity, refraction (surfaces.py:456-520).  `between` runs
// after the intersection and before the refraction (the trace loop's prefetch of the next surface).
template <bool FWD, class M, class P, class F, class S>
__device__ __forceinline__ uint32_t curved_reaction(const S& s, const P& pol, Ray& r, int trips, F between)
{
    uint32_t mask = 0;
    float t;
    // k > -1 (every sphere, ellipsoid, mild asphere) and k <= -1 differ only in the domain test
    // (surfaces.py:727-743); the branch is wave-uniform and taken once, outside the loop
    // (and, for pure conics, a copy for 1 + k == 1: every sphere)
    bool vn;
    if (std::is_same<P, NoPoly>::value && s.unit_k())
        vn = newton_k<M, true, P, true>(s, pol, r, trips, t, mask);
    else
        vn = s.k_gt_m1() ? newton_k<M, true>(s, pol, r, trips, t, mask)
                         : newton_k<M, false>(s, pol, r, trips, t, mask);
    between();
    const float nx = r.ox + t * r.dx, ny = r.oy + t * r.dy, nz = r.oz + t * r.dz;
    bool v;
    if (s.kind() == 1) {                                                                  // :464
        v = nx * nx + ny * ny <= s.r2_lim() && t >= 0.0f && r.ra > 0.0f;
    } else {
        v = vn;                                                                           // :495
    }
    r.ox = v ? nx : r.ox; r.oy = v ? ny : r.oy; r.oz = v ? nz : r.oz;
    r.ra = M::keep_if(r.ra, v);
    refract<FWD, M>(s, pol, r);
    return mask;
}

// Aspheric.ray_reaction, surfaces.py:391-520, on surface `blk` whose constants `s` are already in
// registers.  Returns the Newton convergence mask of this wave on this surface (0 for planes).
template <bool FWD, class M, class F, class S>
__device__ __forceinline__ uint32_t surface_reaction(const S& s, const DevSurface* __restrict__ blk,
                                                     int trips, Ray& r, F between)
{
    if (s.kind() == 0) {
        const float t = M::div(s.d() - r.oz, r.dz);
        const float nx = r.ox + t * r.dx, ny = r.oy + t * r.dy, nz = r.oz + t * r.dz;
        const bool v = M::sqrt(nx * nx + ny * ny) <= s.r_lim() && r.ra > 0.0f;
        r.ox = v ? nx : r.ox; r.oy = v ? ny : r.oy; r.oz = v ? nz : r.oz;
        r.ra = M::keep_if(r.ra, v);
        between();
        if (s.do_refract()) refract<FWD, M>(s, NoPoly{}, r);
        return 0;
    }
    if (s.ai_degree() > 0)                        // wave-uniform: the polynomial block is fetched
        return curved_reaction<FWD, M>(s, s.poly(blk), r, trips, between);   // (one more round trip) on aspheres only
    return curved_reaction<FWD, M>(s, NoPoly{}, r, trips, between);
}

// Ray.propagate_to, basics.py:256-264
template <class M>
__device__ __forceinline__ void propagate_to(Ray& r, float z)
{
    const float t = M::div(z - r.oz, r.dz);
    r.ox = r.ox + r.dx * t;
    r.oy = r.oy + r.dy * t;
    r.oz = r.oz + r.dz * t;
}

// ---- dual-pixel closed-form sub-pixel weights -------------------------------
__device__ __forceinline__ float seg(float u)   // u - 1/2*sin(2u), monte_carlo.py:182
{
    return u - 0.5f * __ocml_sin_f32(2.0f * u);
}

// seg(acos(z)) = acos(z) - z*sqrt(1-z^2), the segment-area function of monte_carlo.py:179-183 (u - sin(2u)/2 at
// u = acos z), for a clamped z in [-1, 1].  Evaluated as ONE function instead of acos + sqrt + product +
// difference: S(z) = (1 - z)^(3/2) G(z) on [0, 1] with G analytic (degree-7 polynomial in u = 2z - 1, fit error
// 4e-9), S(-z) = pi - S(z).  No cancellation next to the clamped ends (S -> 0 there, exactly 0 / pi at z = +-1);
// absolute error <= 3.1e-7 over [0, 1] (2e6 points, against float64), relative error <= 2.7e-7 -- the fp32
// "acos(z) - z sqrt((1-z)(1+z))" of rounds 1-2 had 1.4e-7 absolute and an unbounded relative error, the reference's
// own "u - 0.5 sin(2u)" on MKL acos/sin is of the same order.  It only scales splat WEIGHTS (not bit-pinned:
// DESIGN.md §4): 14 + v_sqrt instructions instead of 23 + two transcendentals, six times per splatted ray.
__device__ __forceinline__ float seg_acos(float z)
{
    const float az = __builtin_fabsf(z);
    const float u = __builtin_fmaf(2.0f, az, -1.0f);
    const float t = 1.0f - az;
    const float w = t * __builtin_amdgcn_sqrtf(t);
    float g = 1.992103051e-06f;
    g = __builtin_fmaf(g, u, -8.485991006e-06f);
    g = __builtin_fmaf(g, u, 3.521309908e-05f);
    g = __builtin_fmaf(g, u, -1.819916826e-04f);
    g = __builtin_fmaf(g, u, 1.097788541e-03f);
    g = __builtin_fmaf(g, u, -8.779404592e-03f);
    g = __builtin_fmaf(g, u, 1.562758839e-01f);
    g = __builtin_fmaf(g, u, 1.737177091e+00f);
    const float sp = w * g;
    return z < 0.0f ? (float)3.141592653589793 - sp : sp;
}

// x / r for the microlens radius: exact multiply when r is a power of two (default 0.5)
__device__ __forceinline__ float over_r(const DevDpParams& p, float x)
{
    return p.r_pow2 ? x * p.inv_r : x / p.r;
}

// monte_carlo.py:169-206 (r <= 0.5); div_fmh(a) = a / fmh.
// M = Lean: the six segment areas through seg_acos (one fused polynomial each).  M = Ieee (SDIRT_PSF_STRICT_IEEE):
// the reference's LITERAL sequence -- torch.clamp, arccos, u - 1/2 sin(2u) -- with ocml's acos / sin, as
// dp_weights_big has it: the on-GPU yardstick of the polynomial (tests/test_gpu_parity.py::
// test_lean_and_literal_subpixel_weights_on_random_geometries).
template <class M, class Div>
__device__ __forceinline__ void dp_weights_small(const DevDpParams& p, const Div& div_fmh, float x_tan,
                                                 float& sl, float& sr)
{
    const float r = p.r, rr = p.rr;
    const float fx = p.f * x_tan;
    float xr = p.w - div_fmh((fx - p.w) * p.h);
    float xm = div_fmh((-fx) * p.h);
    float xl = (-p.w) - div_fmh((fx + p.w) * p.h);
    const float hx = p.h * x_tan;
    if (!M::kFused) {
        xr = clampf(xr, -r, r); xm = clampf(xm, -r, r); xl = clampf(xl, -r, r);
        float ur = __ocml_acos_f32(over_r(p, xr)), um = __ocml_acos_f32(over_r(p, xm)), ul = __ocml_acos_f32(over_r(p, xl));
        float sm = seg(um);
        const float sr_ml = rr * (sm - seg(ur));
        const float sl_ml = rr * (seg(ul) - sm);
        xr = p.w - hx; xm = 0.0f - hx; xl = (-p.w) - hx;
        xr = clampf(xr, -0.5f, 0.5f); xm = clampf(xm, -0.5f, 0.5f); xl = clampf(xl, -0.5f, 0.5f);
        const float xri = clampf(xr, -r, r), xmi = clampf(xm, -r, r), xli = clampf(xl, -r, r);
        ur = __ocml_acos_f32(over_r(p, xri)); um = __ocml_acos_f32(over_r(p, xmi)); ul = __ocml_acos_f32(over_r(p, xli));
        sm = seg(um);
        const float sr_in = rr * (sm - seg(ur));
        const float sl_in = rr * (seg(ul) - sm);
        const float sr_mg = (xr - xm) * 1.0f - sr_in;
        const float sl_mg = (xm - xl) * 1.0f - sl_in;
        sr = sr_ml + sr_mg;
        sl = sl_ml + sl_mg;
        return;
    }
    xr = clamp_finite(xr, -r, r); xm = clamp_finite(xm, -r, r); xl = clamp_finite(xl, -r, r);
    float sm = seg_acos(over_r(p, xm));
    const float sr_ml = rr * (sm - seg_acos(over_r(p, xr)));
    const float sl_ml = rr * (seg_acos(over_r(p, xl)) - sm);
    xr = p.w - hx; xm = 0.0f - hx; xl = (-p.w) - hx;
    xr = clamp_finite(xr, -0.5f, 0.5f); xm = clamp_finite(xm, -0.5f, 0.5f); xl = clamp_finite(xl, -0.5f, 0.5f);
    const float xri = clamp_finite(xr, -r, r), xmi = clamp_finite(xm, -r, r), xli = clamp_finite(xl, -r, r);
    sm = seg_acos(over_r(p, xmi));
    const float sr_in = rr * (sm - seg_acos(over_r(p, xri)));
    const float sl_in = rr * (seg_acos(over_r(p, xli)) - sm);
    const float sr_mg = (xr - xm) * 1.0f - sr_in;
    const float sl_mg = (xm - xl) * 1.0f - sl_in;
    sr = sr_ml + sr_mg;
    sl = sl_ml + sl_mg;
}

__device__ __forceinline__ void dp_weights_small(const DevDpParams& p, float x_tan, float& sl, float& sr)
{
    dp_weights_small<Lean>(p, UDiv<Ieee>::make(p.fmh), x_tan, sl, sr);
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

// UNIT_W: the ray's weight is exactly 0 or 1 (every TRACED ray's is: ra starts at 1 and is only ever
// multiplied by 0/1 validity flags) -- then w is 0 or 1 and the seven products with it are identities.
template <bool UNIT_W = false, class DivY, class DivX>
__device__ __forceinline__ bool splat_taps(const SplatGeom& gm, const DivY& div_dy, const DivX& div_dx,
                                           float sx, float sy, float cx, float cy, float ra, SplatTaps& tp)
{
    float px = (-sx) - cx;                     // points = -o.xy ; points - pointc_ref
    float py = (-sy) - cy;
    float w;
    if (UNIT_W) {
        if (!(ra != 0.0f && __builtin_fabsf(px) < gm.lim && __builtin_fabsf(py) < gm.lim)) return false;
        w = 1.0f;                              // the compiler folds the products below away
    } else {
        w = ra * (__builtin_fabsf(px) < gm.lim ? 1.0f : 0.0f);
        w = w * (__builtin_fabsf(py) < gm.lim ? 1.0f : 0.0f);
        if (!(w != 0.0f)) return false;        // adds exact zeros in the reference
    }
    px = px * w; py = py * w;                  // :38
    const float pf0 = div_dy(py - gm.y_max) * gm.ksm1;   // row
    const float pf1 = div_dx(px - gm.x_min) * gm.ksm1;   // col
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

__device__ __forceinline__ bool splat_taps(const SplatGeom& gm, float sx, float sy, float cx, float cy,
                                           float ra, SplatTaps& tp)
{
    return splat_taps(gm, UDiv<Ieee>::make(gm.dy_rng), UDiv<Ieee>::make(gm.dx_rng), sx, sy, cx, cy, ra, tp);
}

}  // namespace sdirt
