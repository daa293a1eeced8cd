// sdirt_render.hip -- per-pixel PSF convolution kernels of libsdirt_dp.so (MI355X / gfx950 only):
// local_psf_render / local_psf_render_fast / local_dp_psf_render (deeplens/render_psf.py:76-188)
// and PSFNet.pred + render fused over the network's raw outputs (deeplens/psfnet.py:317-336).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/sdirt_dp.h"
#include "sdirt_device.hpp"
#include "sdirt_host.hpp"

using namespace sdirt;

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
// (Round 2's first version moved every weight through LDS -- 16-byte loads, a write, a read per
// weight, two barriers per 8 pixels -- and spent 165 instructions per pixel and wave.)  Here lane l
// loads the weights of ITS taps (f = 64 it + l: consecutive lanes, consecutive floats -- each load
// instruction of a wave is one contiguous 256 bytes of the pixel's kernel) one pixel ahead of the
// one being convolved; LDS holds only the image patch of the workgroup's CHUNK-pixel stretch of
// the row ([KS][CHUNK + KS - 1] positions x 4 channel slots: one 8- or 16-byte read per tap gives
// all channels), staged once: one barrier per workgroup, none around the weights.  The six sums
// of a pixel are reduced together (wave_sum6) and stored by 4 + 2 lanes in two instructions.
template <int C, bool HALF, int KS, int CHUNK>
__global__ void __launch_bounds__(kBlock)
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
            l[it] = __builtin_nontemporal_load(k0 + f);
            r[it] = __builtin_nontemporal_load(k0 + kk + f);
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

// k_psfnet_render with the structure of k_local_psf_render_wave: one wave per pixel, lane l loads the
// raw network outputs of ITS taps (left: f = 64 it + l; right: the fliplr'ed tap, psfnet.py:330 --
// a permutation of the taps, so its values also make up the right kernel's sum) straight from HBM
// one pixel ahead, the image patch of the workgroup's CHUNK pixels sits in LDS with the three
// channels of a position in one 8-byte slot, both normalising sums are reduced together and the
// six outputs with wave_sum6.  Results equal k_psfnet_render's up to the order of the fp32 sums.
template <int C, int KS, int CHUNK>
__global__ void __launch_bounds__(kBlock)
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
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

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
    // wave-per-pixel kernel (ks 21, RGB): one workgroup per 64-pixel stretch of a row; LDS = the
    // stretch's image patch, [21][64 + 20] positions x 4 channel slots (14 KB fp16 / 28 KB fp32)
    constexpr int kChunk = 64;
    const dim3 grid_w((unsigned)((W + kChunk - 1) / kChunk), (unsigned)(B * H));
    auto lds_wave = [&](bool hf) { return (size_t)21 * (kChunk + 20) * 4 * (hf ? 2 : 4); };
#define SDIRT_RENDER_T(CC, HF, PP)                                                               \
    do {                                                                                         \
        if (lds_tile > 48 * 1024)                                                                \
            if (int rc_ = allow_large_lds<&k_local_psf_render_rows<CC, HF, PP, 0>>(64 * 1024)) return rc_; \
        k_local_psf_render_rows<CC, HF, PP, 0><<<grid_t, kBlock, lds_tile, st>>>(                \
            img, psf, H, W, ks, out_l, out_r);                                                   \
    } while (0)
#define SDIRT_RENDER_H(CC, HF)                                                                   \
    do {                                                                                         \
        /* the reference's PSFNet kernel size (configs/dfdp_by_sdirt_rf50mm.yml: ks 21) on RGB */ \
        if (pix == 8 && ks == 21 && CC == 3)                                                     \
            k_local_psf_render_wave<3, HF, 21, kChunk><<<grid_w, kBlock, lds_wave(HF), st>>>(    \
                img, psf, H, W, out_l, out_r);                                                   \
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
#define SDIRT_PN(CC, PP, KK)                                                                     \
    do {                                                                                         \
        const size_t lds = per_pixel * PP;                                                       \
        const int grid = (int)std::min<int64_t>((P + PP - 1) / PP, 256 * 64);                    \
        if (lds > 48 * 1024)                                                                     \
            if (int rc_ = allow_large_lds<&k_psfnet_render<CC, PP, KK>>(64 * 1024)) return rc_;  \
        k_psfnet_render<CC, PP, KK><<<grid, kBlock, lds, st>>>(img, rl, rr, B, H, W, ks, out_l,  \
                                                               out_r);                           \
    } while (0)
#define SDIRT_PN_C(CC)                                                                           \
    do {                                                                                         \
        if (ks == 21 && CC == 3 && (int64_t)3 * H * W < (1ll << 30) && (int64_t)B * H < 65536) { \
            k_psfnet_render_wave<3, 21, 64><<<dim3((unsigned)((W + 63) / 64), (unsigned)(B * H)), kBlock, \
                                              (size_t)21 * 84 * 8, st>>>(img, rl, rr, H, W, out_l, out_r); \
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
