// sdirt_mlp.hip -- the PSF network (deeplens/psfnet_arch.py:26-50) evaluated in ONE kernel.
//
// PSFNet.render (psfnet.py:642-714) runs the MLP 3 -> h/4 -> h -> (h -> h) x L -> ks*ks on
// every pixel of the frame, twice (left: (x,y,z); right: (-x,y,z)).  Layer by layer (stock
// GEMMs) that is HBM-bound: each of the 11 layers streams the [2*H*W, 512] fp16 activations
// out and back in (32 GB per 512x768 frame).  Here a workgroup owns 128 rows of that matrix
// and keeps their activations in LDS for the whole network:
//
//   X  (LDS, 128 rows x 512 features fp16, row stride 1040 B -> conflict-free ds_read_b128)
//   wave w of 4 computes output features [128 w, 128 w + 128) of every layer for all 128 rows
//   with v_mfma_f32_32x32x16_f16 (A = weights, B = X^T), 32 features x 128 rows at a time;
//   A fragments come straight from L2 (weights pre-packed in fragment order: one contiguous
//   1 KB block per (32 outputs x 16 inputs) tile, 16 B per lane), B fragments from LDS;
//   per layer: bias as the C operand, ReLU, round to fp16 (what autocast's Linear + ReLU
//   produce) hidden behind the next group's MFMAs, barrier, overwrite X in place, barrier.
//
// HBM traffic per row: 12 B in, ks*ks*2 B out.  Weights are re-read from L2 once per 128 rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "sdirt_host.hpp"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kRows = 128;            // rows of the activation matrix per workgroup
constexpr int kHid = 512;             // hidden width the kernel is built for
constexpr int kXStride = kHid + 8;    // halves per LDS row: 1040 B, 65 x 16 B
constexpr int kThreads = 256;         // 4 waves, one per SIMD (>= 264 registers each)
constexpr int kMaxLayers = 16;

// Packed network: for every layer its weight fragments [out_pad/32][in_pad/16][64 lanes][8]
// (out_pad = 512, or 128 for the first layer; in_pad = in rounded up to 16), one layer after
// another, then every layer's bias as fp32 [out_pad].  The offsets follow from the widths alone.
struct MlpShape {
    int32_t n_layers;                 // >= 3: 3 -> h4, h4 -> 512, (512 -> 512) x L, 512 -> out
    int32_t h4;                       // width of the first hidden layer (32, 64, 96 or 128)
    int32_t out_features;             // <= 512
};

__host__ __device__ inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

__device__ __forceinline__ f16v mfma(h8 a, h8 b, f16v c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// ReLU of two fp32 values rounded to fp16 (RNE), as one packed register.  max after the
// rounding equals rounding after the max; on the packed pair it is one instruction for two values.
__device__ __forceinline__ h2 relu_pack(float lo, float hi)
{
    h2 v = {(_Float16)lo, (_Float16)hi};
    h2 r;
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}

// One layer for one wave: output features [out_col0, out_col0 + 32 MT) of all 128 rows.
//   wfrag  this wave's first weight tile, [MT][KS][64 lanes] fragments;  KS  k-steps of 16 inputs
// The wave's MT output tiles m are computed ONE AFTER THE OTHER (4 accumulators of 32 features x
// 128 rows at a time) as a single stream of MT * KS steps t = KS m + ks: the weight fragments are
// consecutive in memory in exactly that order (D of them in flight), the X fragments repeat every
// KS steps.  While group m multiplies, the finished accumulators of group m - 1 are read back,
// rounded to fp16 and clamped, so that this arithmetic hides behind MFMAs; only the last group's
// conversion is exposed.  The bias is the C operand of a group's first MFMAs.
//
// Issue order is written out by hand and pinned (sched_barrier after every piece): one step is
//   MFMA n=0 | weight fetch for step t+D (+ one bias quad of the next group near a group's end)
//   MFMA n=1 | X fragments n = 0, 1 of step t+BD
//   MFMA n=2 | X fragments n = 2, 3 of step t+BD
//   MFMA n=3 | half a quad of the previous group's conversion (4 VALU)
// An MFMA occupies the matrix pipe for 32 cycles but the issue port for 8: up to ~5 cheap
// instructions fit in its shadow.  Bunched at the step boundary instead (what the compiler does
// when left alone, or when only the loads are fenced) the same instructions stretch one gap in
// four to 80-100 cycles and the pipe idles a third of the time.
// Fully unrolled: every register index is static.
//   flat_of > 0 (last layer): the results go to X as the dense row-major image [128][flat_of]
//   that the output tile has in global memory, so that it leaves with aligned 16-byte copies.
//   wnext (optional): where this wave's weight stream continues in the NEXT layer; its first
//   fragments are fetched during this layer's last steps, into the ring that the next call
//   receives primed, so that a layer does not start by waiting out an L2 round trip.
#define SDIRT_PIN() __builtin_amdgcn_sched_barrier(0)
template <int KS, int MT, int AD = 6, int BD = 2>
__device__ __forceinline__ void layer(const h8* __restrict__ wfrag, const float* __restrict__ bias,
                                      _Float16* X, int lane, int out_col0, int flat_of = 0,
                                      const h8* __restrict__ wnext = nullptr, h8* ring = nullptr,
                                      bool primed = false)
{
    constexpr int T = MT * KS, D = T < AD ? T : AD;
    constexpr int BR = BD + 1;                           // ring of X fragment sets
    constexpr bool OVERLAP = KS >= 32;                   // 32 half-quads over a group's steps
    const int r = lane & 31, h = lane >> 5;
    const h8* wl = wfrag + lane;
    const _Float16* xl = X + r * kXStride + 8 * h;
    const float* bl = bias + 4 * h;
    h2 packed[MT][4][4][2];                              // [m][n][g] -> 4 halves
    f16v acc[4], done[4];
    f4v bq[4];                                           // bias of the group about to start
    h8 a[D], b[BR][4];
    if (primed) {
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] = ring[d];
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] = wl[d * 64];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f4v*>(bl + 8 * g);
#pragma unroll
    for (int d = 0; d < BD; ++d)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (d < T) b[d][n] = *reinterpret_cast<const h8*>(xl + 32 * n * kXStride + 16 * (d % KS));
    SDIRT_PIN();
#pragma clang loop unroll(full)
    for (int t = 0; t < T; ++t) {
        const int m = t / KS, ks = t % KS;
        const h8 at = a[t % D];
        f16v c0[4];
        if (ks == 0) {
            // accumulator register 4 g + i of a lane holds output row 8 g + 4 h + i of the tile
            f16v bv;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[4 * g + i] = bq[g][i];
#pragma unroll
            for (int n = 0; n < 4; ++n) c0[n] = bv;
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n) c0[n] = acc[n];
        }
        // ---- MFMA 0 | weights (and bias) ----
        acc[0] = mfma(at, b[t % BR][0], c0[0]);
        if (t + D < T) a[t % D] = wl[(t + D) * 64];
        else if (wnext) a[t % D] = wnext[lane + (t + D - T) * 64];
        if (m + 1 < MT && KS >= 4 && ks >= KS - 4) {
            const int g = ks - (KS - 4);
            bq[g] = *reinterpret_cast<const f4v*>(bl + 32 * (m + 1) + 8 * g);
        } else if (m + 1 < MT && KS < 4 && ks == KS - 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f4v*>(bl + 32 * (m + 1) + 8 * g);
        }
        SDIRT_PIN();
        // ---- MFMA 1 | X fragments 0, 1 ----
        acc[1] = mfma(at, b[t % BR][1], c0[1]);
        if (t + BD < T) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
                b[(t + BD) % BR][n] =
                    *reinterpret_cast<const h8*>(xl + 32 * n * kXStride + 16 * ((t + BD) % KS));
        }
        SDIRT_PIN();
        // ---- MFMA 2 | X fragments 2, 3 ----
        acc[2] = mfma(at, b[t % BR][2], c0[2]);
        if (t + BD < T) {
#pragma unroll
            for (int n = 2; n < 4; ++n)
                b[(t + BD) % BR][n] =
                    *reinterpret_cast<const h8*>(xl + 32 * n * kXStride + 16 * ((t + BD) % KS));
        }
        SDIRT_PIN();
        // ---- MFMA 3 | half a quad of the previous group ----
        acc[3] = mfma(at, b[t % BR][3], c0[3]);
        if (OVERLAP && m > 0) {
            const int q = ks / 2, n = q / 4, g = q % 4, half = ks & 1;
            packed[m - 1][n][g][half] = relu_pack(done[n][4 * g + 2 * half], done[n][4 * g + 2 * half + 1]);
        }
        SDIRT_PIN();
        if (ks == KS - 1) {
#pragma unroll
            for (int n = 0; n < 4; ++n) done[n] = acc[n];
            if (m == MT - 1 || !OVERLAP) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        packed[m][n][g][0] = relu_pack(done[n][4 * g + 0], done[n][4 * g + 1]);
                        packed[m][n][g][1] = relu_pack(done[n][4 * g + 2], done[n][4 * g + 3]);
                    }
            }
        }
    }
    if (ring) {
        // step T + d of the stream sits in slot (T + d) % D; hand the ring over in step order
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = a[(T + d) % D];
    }
    __syncthreads();                       // every wave has finished reading X
    if (flat_of > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = out_col0 + 32 * m + 8 * g + 4 * h;
                    _Float16* dst = X + (32 * n + r) * flat_of + f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (f + i < flat_of) dst[i] = packed[m][n][g][i >> 1][i & 1];
                }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    struct alignas(8) Pair { h2 lo, hi; } pr{packed[m][n][g][0], packed[m][n][g][1]};
                    *reinterpret_cast<Pair*>(X + (32 * n + r) * kXStride + out_col0 + 32 * m + 8 * g + 4 * h) = pr;
                }
    }
    __syncthreads();
}
#undef SDIRT_PIN

template <int AD, int BD>
__global__ void __launch_bounds__(kThreads, 1)
k_psfnet_mlp(MlpShape shape, const h8* __restrict__ wpk, const float* __restrict__ bpk,
             const float* __restrict__ inp, int64_t n_points, int mirror, _Float16* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 X[];      // [kRows][kXStride]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_rows = n_points * (mirror ? 2 : 1);
    const int64_t n_tiles = (n_rows + kRows - 1) / kRows;
    const int of = shape.out_features, h4 = shape.h4;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * kRows;
        // layer-0 input: (x, y, z) as fp16 in features 0..2, zeros up to 16
        for (int i = threadIdx.x; i < kRows * 2; i += kThreads) {
            const int row = i >> 1, half = i & 1;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            const int64_t gr = row0 + row;
            if (half == 0 && gr < n_rows) {
                const bool right = gr >= n_points;
                const float* p = inp + (right ? gr - n_points : gr) * 3;
                v[0] = (_Float16)(right ? -p[0] : p[0]);
                v[1] = (_Float16)p[1];
                v[2] = (_Float16)p[2];
            }
            *reinterpret_cast<h8*>(X + row * kXStride + 8 * half) = v;
        }
        __syncthreads();
        const h8* w = wpk;
        const float* b = bpk;
        // 3 -> h4: one 32-feature tile per wave while h4 allows
        if (32 * wave < h4) layer<1, 1>(w + wave * 64, b + 32 * wave, X, lane, 32 * wave);
        else { __syncthreads(); __syncthreads(); }
        w += 128 / 32 * 64;                       // first layer: 128 rows x 16 columns
        b += 128;
        // h4 -> 512, (512 -> 512) x L, 512 -> out (rows zero-padded to 512)
        int ksn = h4 / 16;
        h8 ring[AD];
        bool primed = false;
        for (int l = 1; l < shape.n_layers; ++l) {
            const h8* wt = w + 4 * wave * ksn * 64;
            const float* bt = b + 128 * wave;
            const bool last = l + 1 == shape.n_layers;
            // the next layer's stream of this wave (a wave idle in the last layer skips it)
            const h8* wn = (!last && !(l + 2 == shape.n_layers && 128 * wave >= of))
                               ? w + 16 * ksn * 64 + 4 * wave * 32 * 64 : nullptr;
            if (last && 128 * wave >= of) { __syncthreads(); __syncthreads(); }
            else if (ksn == 32) {
                layer<32, 4, AD, BD>(wt, bt, X, lane, 128 * wave, last ? of : 0, wn, ring, primed);
                primed = wn != nullptr;
            }
            else if (ksn == 8) layer<8, 4>(wt, bt, X, lane, 128 * wave);
            else if (ksn == 6) layer<6, 4>(wt, bt, X, lane, 128 * wave);
            else if (ksn == 4) layer<4, 4>(wt, bt, X, lane, 128 * wave);
            else layer<2, 4>(wt, bt, X, lane, 128 * wave);
            w += 16 * ksn * 64;                   // 512 output rows of this layer
            b += kHid;
            ksn = kHid / 16;
        }
        // X now holds the tile as it lies in global memory: rows * of contiguous halves, starting
        // 16-byte aligned because row0 is a multiple of 128 and the host checks `out`
        const int64_t valid = std::min<int64_t>(kRows, n_rows - row0);
        const int total = (int)valid * of;
        _Float16* dst = out + row0 * of;
        for (int i0 = threadIdx.x * 8; i0 < total; i0 += kThreads * 8) {
            const h8 v = *reinterpret_cast<const h8*>(X + i0);
            if (i0 + 8 <= total) *reinterpret_cast<h8*>(dst + i0) = v;
            else
                for (int j = 0; i0 + j < total; ++j) dst[i0 + j] = v[j];
        }
        __syncthreads();
    }
}

__global__ void k_bias_pad(const float* __restrict__ bias, int out_f, int out_pad, float* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < out_pad) dst[i] = i < out_f ? bias[i] : 0.0f;
}

// fp32 [out, in] row-major -> fp16 fragments [out_pad/32][in_pad/16][64][8]:
// lane (r = l & 31, h = l >> 5) of tile (mt, ks) holds W[32 mt + r][16 ks + 8 h + j], j < 8.
__global__ void k_mlp_pack(const float* __restrict__ w, int out_f, int in_f, int out_pad, int in_pad,
                           _Float16* __restrict__ packed)
{
    const int64_t total = (int64_t)out_pad * in_pad;
    const int ksn = in_pad / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t t = i >> 9;
        const int ks = (int)(t % ksn), mt = (int)(t / ksn);
        const int row = 32 * mt + (lane & 31), col = 16 * ks + 8 * (lane >> 5) + j;
        packed[i] = (row < out_f && col < in_f) ? (_Float16)w[(int64_t)row * in_f + col] : (_Float16)0;
    }
}

int check_widths(const int32_t* widths, int32_t n_layers)
{
    if (!widths) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null widths");
    if (n_layers < 3 || n_layers > kMaxLayers)
        return fail(SDIRT_ERR_UNSUPPORTED, "n_layers=%d outside [3,%d]", n_layers, kMaxLayers);
    const int h4 = widths[1];
    if (widths[0] != 3 || h4 % 32 != 0 || h4 < 32 || h4 > 128)
        return fail(SDIRT_ERR_UNSUPPORTED, "first layer must be 3 -> 32/64/96/128 (got %d -> %d)",
                    widths[0], h4);
    for (int l = 2; l < n_layers; ++l)
        if (widths[l] != kHid)
            return fail(SDIRT_ERR_UNSUPPORTED, "hidden width %d at layer %d; the kernel is built for %d",
                        widths[l], l, kHid);
    if (widths[n_layers] < 1 || widths[n_layers] > kHid)
        return fail(SDIRT_ERR_UNSUPPORTED, "out_features=%d outside [1,%d]", widths[n_layers], kHid);
    return SDIRT_OK;
}

// padded output rows of layer l: 128 for the first layer, 512 for every other one
inline int out_rows(int l) { return l == 0 ? 128 : kHid; }

// bytes of all weight fragments; the biases follow
int64_t weights_bytes(const int32_t* widths, int32_t n_layers)
{
    int64_t halves = 0;
    for (int l = 0; l < n_layers; ++l) halves += (int64_t)out_rows(l) * pad_to(widths[l], 16);
    return halves * (int64_t)sizeof(_Float16);
}

}  // namespace

extern "C" {

int64_t sdirt_mlp_packed_bytes(const int32_t* widths, int32_t n_layers)
{
    if (check_widths(widths, n_layers) != SDIRT_OK) return -1;
    return weights_bytes(widths, n_layers) +
           (int64_t)sizeof(float) * (128 + (int64_t)kHid * (n_layers - 1));
}

int sdirt_mlp_pack(const float* const* weights, const float* const* biases, const int32_t* widths,
                   int32_t n_layers, void* packed, void* stream)
{
    if (!weights || !biases || !packed) return fail(SDIRT_ERR_INVALID_ARGUMENT, "null argument");
    if ((uintptr_t)packed & 15) return fail(SDIRT_ERR_INVALID_ARGUMENT, "packed must be 16-byte aligned");
    const int rc = check_widths(widths, n_layers);
    if (rc != SDIRT_OK) return rc;
    hipStream_t st = as_stream(stream);
    _Float16* w = static_cast<_Float16*>(packed);
    float* b = reinterpret_cast<float*>(static_cast<char*>(packed) + weights_bytes(widths, n_layers));
    for (int l = 0; l < n_layers; ++l) {
        if (!weights[l] || !biases[l]) return fail(SDIRT_ERR_INVALID_ARGUMENT, "layer %d: null pointer", l);
        const int out_f = widths[l + 1], in_f = widths[l];
        const int op = out_rows(l), ip = pad_to(in_f, 16);
        const int64_t total = (int64_t)op * ip;
        k_mlp_pack<<<(int)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, st>>>(
            weights[l], out_f, in_f, op, ip, w);
        k_bias_pad<<<(op + 255) / 256, 256, 0, st>>>(biases[l], out_f, op, b);
        w += total;
        b += op;
    }
    LAUNCH_CHECK();
    return SDIRT_OK;
}

int sdirt_psfnet_mlp(const void* packed, const int32_t* widths, int32_t n_layers, const float* inp,
                     int64_t n_points, int32_t mirror, void* out, void* stream)
{
    if (!packed || !inp || !out || n_points < 0) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    const int rc = check_widths(widths, n_layers);
    if (rc != SDIRT_OK) return rc;
    if (((uintptr_t)out | (uintptr_t)packed) & 15)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "packed and out must be 16-byte aligned");
    if (n_points == 0) return SDIRT_OK;
    MlpShape shape{n_layers, widths[1], widths[n_layers]};
    const h8* w = static_cast<const h8*>(packed);
    const float* b = reinterpret_cast<const float*>(static_cast<const char*>(packed) +
                                                    weights_bytes(widths, n_layers));
    const size_t lds = sizeof(_Float16) * kRows * kXStride;           // 133,120 B
#ifndef SDIRT_MLP_AD
#define SDIRT_MLP_AD 6
#define SDIRT_MLP_BD 2
#endif
    auto kern = k_psfnet_mlp<SDIRT_MLP_AD, SDIRT_MLP_BD>;   // 6 weight fragments, 2 sets of X fragments in flight
    if (int rc_ = allow_large_lds<&k_psfnet_mlp<SDIRT_MLP_AD, SDIRT_MLP_BD>>((int)lds)) return rc_;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t rows = n_points * (mirror ? 2 : 1);
    const int64_t tiles = (rows + kRows - 1) / kRows;
    const int grid = (int)std::min<int64_t>(tiles, cus);
    kern<<<grid, kThreads, lds, as_stream(stream)>>>(shape, w, b, inp, n_points, mirror ? 1 : 0,
                                                            static_cast<_Float16*>(out));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

}  // extern "C"
