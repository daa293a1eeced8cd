// sdirt_hostrng.hip -- host only: torch's CPU random stream, generated faster.
//
// Lensgroup.sample_from_points draws its pupil uniforms with torch.rand on the CPU default generator
// (deeplens/optics.py:483-484) -- that is what makes seeds reproduce, so the numbers are part of the
// contract.  torch generates them one at a time (MT19937, ~1.6 ns each: 70 us for the 44096 numbers of
// a PSFNet fitting batch -- a fifth of that call).  This file produces THE SAME numbers from the same
// generator state a block of 624 at a time with loops the compiler vectorises (AVX2 where the CPU has
// it), and leaves the state exactly where torch would have left it.  The binding checks the
// equivalence against torch.rand itself before it ever uses this (sdirt_amd/_hostrng.py) and falls
// back to torch.rand if anything differs.
#include <stdint.h>
#include <string.h>

#include "../../include/sdirt_dp.h"
#include "sdirt_host.hpp"

namespace {

constexpr int kN = 624, kM = 397;

inline uint32_t twist(uint32_t u, uint32_t v)
{
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

// at/core/MT19937RNGEngine.h: next_state().  old -> nw (separate arrays: every loop below reads only
// `old` and entries of `nw` written by an EARLIER loop, so each one vectorises).
#define SDIRT_RNG_BODY                                                                            \
    for (int i = 0; i < kN - kM; ++i) nw[i] = old[i + kM] ^ twist(old[i], old[i + 1]);           \
    for (int i = kN - kM; i < 2 * (kN - kM); ++i) nw[i] = nw[i - (kN - kM)] ^ twist(old[i], old[i + 1]); \
    for (int i = 2 * (kN - kM); i < kN - 1; ++i) nw[i] = nw[i - (kN - kM)] ^ twist(old[i], old[i + 1]);  \
    nw[kN - 1] = nw[kM - 1] ^ twist(old[kN - 1], nw[0]);

#define SDIRT_RNG_TEMPER                                                                          \
    for (int64_t i = 0; i < cnt; ++i) {                                                           \
        uint32_t y = src[i];                                                                      \
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18); \
        /* at::uniform_real_distribution<float>: (x & (2^24 - 1)) * 2^-24 */                      \
        dst[i] = (float)(y & 0xffffffu) * (1.0f / 16777216.0f);                                   \
    }

__attribute__((target("avx2"))) void next_state_avx2(const uint32_t* __restrict__ old, uint32_t* __restrict__ nw)
{
    SDIRT_RNG_BODY
}
void next_state_base(const uint32_t* __restrict__ old, uint32_t* __restrict__ nw) { SDIRT_RNG_BODY }

__attribute__((target("avx2"))) void temper_avx2(const uint32_t* __restrict__ src, int64_t cnt, float* __restrict__ dst)
{
    SDIRT_RNG_TEMPER
}
void temper_base(const uint32_t* __restrict__ src, int64_t cnt, float* __restrict__ dst) { SDIRT_RNG_TEMPER }

}  // namespace

extern "C" int sdirt_host_uniform_fill(void* th_state, int64_t state_bytes, int64_t n, float* out)
{
    // THGeneratorState as torch.get_rng_state() serialises the CPU generator:
    //   uint64 seed | int32 left | int32 seeded | uint64 next | uint64 state[624] | normal-sample cache
    if (!th_state || !out || n < 0 || state_bytes < 24 + 8 * kN)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    uint8_t* st = static_cast<uint8_t*>(th_state);
    int32_t left;
    uint64_t next;
    memcpy(&left, st + 8, 4);
    memcpy(&next, st + 16, 8);
    if (left < 1 || left > kN + 1 || next > (uint64_t)kN)
        return fail(SDIRT_ERR_UNSUPPORTED, "generator state layout not recognised (left=%d next=%llu)", left,
                    (unsigned long long)next);
    uint32_t a[kN], b[kN];
    uint64_t w;
    for (int i = 0; i < kN; ++i) {
        memcpy(&w, st + 24 + 8 * i, 8);
        if (w >> 32) return fail(SDIRT_ERR_UNSUPPORTED, "generator state layout not recognised (word %d)", i);
        a[i] = (uint32_t)w;
    }
    static const bool avx2 = __builtin_cpu_supports("avx2");
    uint32_t *cur = a, *oth = b;
    while (n > 0) {
        int64_t avail = (int64_t)left - 1;                 // values of this block not handed out yet
        if (avail == 0) {                                  // the engine's `if (--left == 0) next_state()`
            if (avx2) next_state_avx2(cur, oth); else next_state_base(cur, oth);
            uint32_t* t = cur; cur = oth; oth = t;
            left = kN + 1;
            next = 0;
            avail = kN;
        }
        const int64_t cnt = avail < n ? avail : n;
        if (avx2) temper_avx2(cur + next, cnt, out); else temper_base(cur + next, cnt, out);
        out += cnt; n -= cnt;
        left -= (int32_t)cnt;
        next += (uint64_t)cnt;
    }
    memcpy(st + 8, &left, 4);
    memcpy(st + 16, &next, 8);
    for (int i = 0; i < kN; ++i) {
        w = cur[i];
        memcpy(st + 24 + 8 * i, &w, 8);
    }
    return SDIRT_OK;
}
