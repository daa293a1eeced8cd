// sdirt_mlp.hip -- the PSF network (deeplens/psfnet_arch.py:26-50) evaluated in ONE kernel.
//
// PSFNet.render (psfnet.py:642-714) runs the MLP 3 -> h/4 -> h -> (h -> h) x L -> ks*ks on
// every pixel of the frame, twice (left: (x,y,z); right: (-x,y,z)).  Layer by layer (stock
// GEMMs) that is HBM-bound: each of the 11 layers streams the [2*H*W, 512] fp16 activations
// out and back in (32 GB per 512x768 frame).  Here a workgroup owns 128 rows of that matrix
// and keeps their activations in LDS for the whole network:
//
//   X  (LDS, 128 rows x 512 features fp16, row stride 1040 B -> conflict-free ds_read_b128)
//   wave w of 4 computes output features [128 w, 128 w + 128) of every layer for all 128 rows:
//   16 accumulator tiles of v_mfma_f32_32x32x16_f16 (A = weights, B = X^T), 256 registers;
//   A fragments come straight from L2 (weights pre-packed in fragment order: one contiguous
//   1 KB block per (32 outputs x 16 inputs) tile, 16 B per lane), B fragments from LDS;
//   epilogue per layer: + bias, ReLU, round to fp16 (what autocast's Linear + ReLU produce),
//   barrier, overwrite X in place, barrier.
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

// One layer for one wave: features [out_col0, out_col0 + 32 MT) of all 128 rows.
//   wfrag  this wave's first weight tile, [MT][ksn][64 lanes] fragments;  ksn  k-steps of 16 inputs
template <int MT>
__device__ __forceinline__ void layer(const h8* __restrict__ wfrag, const float* __restrict__ bias,
                                      int ksn, _Float16* X, int lane, int out_col0)
{
    const int r = lane & 31, h = lane >> 5;
    f16v acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // accumulator register 4 g + i of a lane holds output row 8 g + 4 h + i of the tile
            const f4v bv = *reinterpret_cast<const f4v*>(bias + 32 * m + 8 * g + 4 * h);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[m][n][4 * g + i] = bv[i];
        }
    const h8* wl = wfrag + lane;
    const _Float16* xl = X + r * kXStride + 8 * h;
    const int mstride = ksn * 64;
    h8 a0[MT], a1[MT], b0[4], b1[4];
#pragma unroll
    for (int m = 0; m < MT; ++m) a0[m] = wl[m * mstride];
#pragma unroll
    for (int n = 0; n < 4; ++n) b0[n] = *reinterpret_cast<const h8*>(xl + 32 * n * kXStride);
    // two k-steps per trip, the operands of one fetched while the other multiplies; straight-line
    // bodies only (a branch around an MFMA group makes the compiler shuttle all 256
    // accumulators between register files at every trip).  ksn is even (MT == 4) or 1 (MT == 1).
#define SDIRT_LOAD(A, B, KSTEP)                                                                  \
    do {                                                                                         \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) A[m] = wl[m * mstride + (KSTEP) * 64];    \
        _Pragma("unroll") for (int n = 0; n < 4; ++n)                                            \
            B[n] = *reinterpret_cast<const h8*>(xl + 32 * n * kXStride + 16 * (KSTEP));          \
        /* keep the fetches of the NEXT step ahead of this step's MFMAs: left alone, the   */    \
        /* scheduler sinks each load to just before its use and waits out the L2 latency */      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define SDIRT_MFMA(A, B)                                                                         \
    do {                                                                                         \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                           \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[m][n] = mfma(A[m], B[n], acc[m][n]); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
    if (MT == 1) {                         // the 3 -> h4 layer: a single k-step
        SDIRT_MFMA(a0, b0);
    } else {
        for (int ks = 0; ks < ksn - 2; ks += 2) {
            SDIRT_LOAD(a1, b1, ks + 1);
            SDIRT_MFMA(a0, b0);
            SDIRT_LOAD(a0, b0, ks + 2);
            SDIRT_MFMA(a1, b1);
        }
        SDIRT_LOAD(a1, b1, ksn - 1);
        SDIRT_MFMA(a0, b0);
        SDIRT_MFMA(a1, b1);
    }
#undef SDIRT_LOAD
#undef SDIRT_MFMA
    __syncthreads();                       // every wave has finished reading X
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = (_Float16)fmaxf(acc[m][n][4 * g + i], 0.0f);
                *reinterpret_cast<h4*>(X + (32 * n + r) * kXStride + out_col0 + 32 * m + 8 * g + 4 * h) = v;
            }
    __syncthreads();
}

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
        if (32 * wave < h4) layer<1>(w + wave * 64, b + 32 * wave, 1, X, lane, 32 * wave);
        else { __syncthreads(); __syncthreads(); }
        w += 128 / 32 * 64;                       // first layer: 128 rows x 16 columns
        b += 128;
        // h4 -> 512, (512 -> 512) x L, 512 -> out (rows zero-padded to 512)
        int ksn = h4 / 16;
        for (int l = 1; l < shape.n_layers; ++l) {
            if (l + 1 < shape.n_layers || 128 * wave < of)
                layer<4>(w + 4 * wave * ksn * 64, b + 128 * wave, ksn, X, lane, 128 * wave);
            else { __syncthreads(); __syncthreads(); }
            w += 16 * ksn * 64;                   // 512 output rows of this layer
            b += kHid;
            ksn = kHid / 16;
        }
        // X[row][0 .. of) -> out[row0 + row][0 .. of): one contiguous run of rows * of halves
        const int64_t valid = std::min<int64_t>(kRows, n_rows - row0);
        const int total = (int)valid * of;
        _Float16* dst = out + row0 * of;
        // 16-byte aligned because row0 is a multiple of 128 and the host checks `out`
        for (int i8 = threadIdx.x; i8 * 8 < total; i8 += kThreads) {
            const int i0 = i8 * 8;
            h8 v;
            int row = i0 / of, col = i0 - row * of;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = (i0 + j < total) ? X[row * kXStride + col] : (_Float16)0;
                if (++col == of) { col = 0; ++row; }
            }
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
    HIP_TRY(hipFuncSetAttribute((const void*)k_psfnet_mlp, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t rows = n_points * (mirror ? 2 : 1);
    const int64_t tiles = (rows + kRows - 1) / kRows;
    const int grid = (int)std::min<int64_t>(tiles, cus);
    k_psfnet_mlp<<<grid, kThreads, lds, as_stream(stream)>>>(shape, w, b, inp, n_points, mirror ? 1 : 0,
                                                            static_cast<_Float16*>(out));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

}  // extern "C"
