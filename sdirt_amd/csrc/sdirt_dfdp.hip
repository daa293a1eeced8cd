// sdirt_dfdp.hip -- the dual-pixel cost volume of the depth-from-DP network
// (dfdp/dddnet/dddnet.py:136-148, 155-178): for every signed shift gap = i - D/2 the left features
// and the right features displaced by gap, side by side on the channel axis, zero where the shift
// leaves the image.  The reference zero-fills [B, 2C, D, H, W] and then issues 2 D sliced copies;
// here every output element is written exactly once, 16 bytes per lane.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "sdirt_host.hpp"

namespace {

template <typename T>
__global__ void __launch_bounds__(256)
k_dp_cost_volume(const T* __restrict__ x, const T* __restrict__ y, int B, int C, int D, int H, int W,
                 T* __restrict__ cost)
{
    // one thread per (b, c2, i, h, w-quad); w fastest so that stores coalesce
    const int wq = (W + 3) / 4;
    const int64_t total = (int64_t)B * 2 * C * D * H * wq;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(t % wq);
        int64_t rest = t / wq;
        const int h = (int)(rest % H); rest /= H;
        const int i = (int)(rest % D); rest /= D;
        const int c2 = (int)(rest % (2 * C));
        const int b = (int)(rest / (2 * C));
        const int gap = i - D / 2;
        const bool right = c2 >= C;
        const T* src = (right ? y : x) + (((int64_t)b * C + (right ? c2 - C : c2)) * H + h) * W;
        T* dst = cost + ((((int64_t)b * 2 * C + c2) * D + i) * H + h) * W;
        const int lo = gap > 0 ? gap : 0, hi = gap < 0 ? W + gap : W;     // columns that receive data
        const int shift = right ? gap : 0;                                // y is read at w - gap
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = 4 * q + j;
            if (w < W) dst[w] = (w >= lo && w < hi) ? src[w - shift] : (T)0;
        }
    }
}

// The same volume for feature maps stored pixel-major (the storage of a channels_last tensor: [B, H, W, C]) into a volume
// stored [B, D, H, W, 2C] (channels_last_3d) -- the layout MIOpen's 3-D convolutions of the hourglass compute in, so that
// nothing is transposed on the way from the feature network to the first of them.  One thread per 16-byte group of
// channels of one (b, i, h, w): a pixel's 2C channels are one contiguous run, loads and stores are both coalesced.
template <typename T, int VL>
__global__ void __launch_bounds__(256)
k_dp_cost_volume_nhwc(const T* __restrict__ x, const T* __restrict__ y, int B, int C, int D, int H, int W,
                      T* __restrict__ cost)
{
    typedef T vec __attribute__((ext_vector_type(VL)));
    const int groups = 2 * C / VL, half = C / VL;
    const int64_t total = (int64_t)B * D * H * W * groups;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        int64_t rest = t / groups;
        const int w = (int)(rest % W); rest /= W;
        const int h = (int)(rest % H); rest /= H;
        const int i = (int)(rest % D);
        const int b = (int)(rest / D);
        const int gap = i - D / 2;
        const bool right = g >= half;
        const int lo = gap > 0 ? gap : 0, hi = gap < 0 ? W + gap : W;     // columns that receive data
        const int ws = right ? w - gap : w;                               // y is read at w - gap
        vec v = {};
        if (w >= lo && w < hi)
            v = *reinterpret_cast<const vec*>((right ? y : x) + (((int64_t)b * H + h) * W + ws) * C +
                                              (int64_t)(right ? g - half : g) * VL);
        *reinterpret_cast<vec*>(cost + t * VL) = v;
    }
}

// Adjoint of the cost volume: every input element collects the gradient of the (up to D) volume
// elements it was copied to.  One thread per input element, reads coalesced along w.
template <typename T>
__global__ void __launch_bounds__(256)
k_dp_cost_volume_bwd(const T* __restrict__ gcost, int B, int C, int D, int H, int W, T* __restrict__ gx,
                     T* __restrict__ gy)
{
    const int64_t total = (int64_t)B * 2 * C * H * W;
    const int64_t plane = (int64_t)H * W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(t % W);
        int64_t rest = t / W;
        const int h = (int)(rest % H); rest /= H;
        const int c2 = (int)(rest % (2 * C));
        const int b = (int)(rest / (2 * C));
        const bool right = c2 >= C;
        const T* src = gcost + (((int64_t)b * 2 * C + c2) * D) * plane + (int64_t)h * W;
        float acc = 0.0f;
        for (int i = 0; i < D; ++i) {
            const int gap = i - D / 2;
            const int wv = right ? w + gap : w;                        // column of the volume it went to
            const int lo = gap > 0 ? gap : 0, hi = gap < 0 ? W + gap : W;
            if (wv >= lo && wv < hi) acc += (float)src[i * plane + wv];
        }
        T* dst = right ? gy : gx;
        dst[(((int64_t)b * C + (right ? c2 - C : c2)) * H + h) * W + w] = (T)acc;
    }
}

}  // namespace

extern "C" int sdirt_dp_cost_volume_backward(const void* grad_cost, int32_t batch, int32_t channels,
                                             int32_t d_max, int32_t height, int32_t width,
                                             int32_t half_precision, void* grad_x, void* grad_y, void* stream)
{
    if (!grad_cost || !grad_x || !grad_y || batch < 0 || channels < 1 || d_max < 1 || height < 1 || width < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (batch == 0) return SDIRT_OK;
    const int64_t total = (int64_t)batch * 2 * channels * height * width;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
    if (half_precision)
        k_dp_cost_volume_bwd<_Float16><<<grid, 256, 0, as_stream(stream)>>>(
            static_cast<const _Float16*>(grad_cost), batch, channels, d_max, height, width,
            static_cast<_Float16*>(grad_x), static_cast<_Float16*>(grad_y));
    else
        k_dp_cost_volume_bwd<float><<<grid, 256, 0, as_stream(stream)>>>(
            static_cast<const float*>(grad_cost), batch, channels, d_max, height, width,
            static_cast<float*>(grad_x), static_cast<float*>(grad_y));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

extern "C" int sdirt_dp_cost_volume(const void* x, const void* y, int32_t batch, int32_t channels,
                                    int32_t d_max, int32_t height, int32_t width, int32_t half_precision,
                                    void* cost, void* stream)
{
    if (!x || !y || !cost || batch < 0 || channels < 1 || d_max < 1 || height < 1 || width < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (d_max / 2 >= width) return fail(SDIRT_ERR_INVALID_ARGUMENT, "d_max/2 must be < width");
    if (batch == 0) return SDIRT_OK;
    const int64_t total = (int64_t)batch * 2 * channels * d_max * height * ((width + 3) / 4);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
    if (half_precision)
        k_dp_cost_volume<_Float16><<<grid, 256, 0, as_stream(stream)>>>(
            static_cast<const _Float16*>(x), static_cast<const _Float16*>(y), batch, channels, d_max, height,
            width, static_cast<_Float16*>(cost));
    else
        k_dp_cost_volume<float><<<grid, 256, 0, as_stream(stream)>>>(
            static_cast<const float*>(x), static_cast<const float*>(y), batch, channels, d_max, height, width,
            static_cast<float*>(cost));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

extern "C" int sdirt_dp_cost_volume_nhwc(const void* x, const void* y, int32_t batch, int32_t channels,
                                         int32_t d_max, int32_t height, int32_t width, int32_t half_precision,
                                         void* cost, void* stream)
{
    if (!x || !y || !cost || batch < 0 || channels < 1 || d_max < 1 || height < 1 || width < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (d_max / 2 >= width) return fail(SDIRT_ERR_INVALID_ARGUMENT, "d_max/2 must be < width");
    if (batch == 0) return SDIRT_OK;
    // 16-byte groups where the channel count and the pointers allow it, single elements otherwise
    const int vl = half_precision ? 8 : 4;
    const bool wide = channels % vl == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)cost) & 15) == 0;
    const int64_t total = (int64_t)batch * d_max * height * width * (2 * channels / (wide ? vl : 1));
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
    hipStream_t st = as_stream(stream);
#define SDIRT_CV(T, VL) k_dp_cost_volume_nhwc<T, VL><<<grid, 256, 0, st>>>(static_cast<const T*>(x), static_cast<const T*>(y), \
                            batch, channels, d_max, height, width, static_cast<T*>(cost))
    if (half_precision) { if (wide) SDIRT_CV(_Float16, 8); else SDIRT_CV(_Float16, 1); }
    else { if (wide) SDIRT_CV(float, 4); else SDIRT_CV(float, 1); }
#undef SDIRT_CV
    LAUNCH_CHECK();
    return SDIRT_OK;
}

// ---------------------------------------------------------------------------
// Inference-time fusions of the depth network (eval mode; training keeps torch's differentiable ops).
// ---------------------------------------------------------------------------

// BasicConv's tail (dddnet.py:539-543): batch norm with its running statistics, then ReLU, in ONE pass over the convolution's
// output, in place (MIOpen's inference batch norm + torch's clamp are two).  The arithmetic is torch's batch_norm transform
// term by term -- gamma * (x - mean) * invstd + beta in fp32, rounded once to the tensor's type.
// The tensor is [outer][C][inner] (planar: inner = the spatial size) or pixel-major (channels_last: inner = 1, C fastest).
template <typename T>
__device__ __forceinline__ T bn_relu_one(T x, float mean, float invstd, float gamma, float beta, int relu)
{
    float y = gamma * ((float)x - mean) * invstd + beta;
    if (relu) y = y < 0.0f ? 0.0f : y;                  // (NaN passes, as clamp_min lets it)
    return (T)y;
}

// pixel-major tensors [pixels][C]: a thread keeps ONE group of VL channels (its 4 x VL constants in registers) and walks
// down the pixels; consecutive threads hold consecutive 16-byte groups, so loads and stores are coalesced.
// blockDim.x = groups * (pixels per block pass), groups = C / VL.
template <typename T, int VL>
__global__ void __launch_bounds__(256)
k_bn_relu_pixel_major(T* __restrict__ x, int64_t n_pixels, int C, const float* __restrict__ mean,
                      const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, int relu)
{
    typedef T vec __attribute__((ext_vector_type(VL)));
    const int groups = C / VL;
    const int g = threadIdx.x % groups, ppb = blockDim.x / groups;
    float m[VL], is[VL], ga[VL], be[VL];
#pragma unroll
    for (int j = 0; j < VL; ++j) {
        m[j] = mean[g * VL + j]; is[j] = invstd[g * VL + j]; ga[j] = gamma[g * VL + j]; be[j] = beta[g * VL + j];
    }
    for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / groups; p < n_pixels; p += (int64_t)gridDim.x * ppb) {
        vec* q = reinterpret_cast<vec*>(x + p * C + g * VL);
        vec v = *q;
#pragma unroll
        for (int j = 0; j < VL; ++j) v[j] = bn_relu_one<T>(v[j], m[j], is[j], ga[j], be[j], relu);
        *q = v;
    }
}

// planar tensors [planes = B x C][inner]: a workgroup stays inside one plane (its four constants are uniform: scalar loads)
template <typename T, int VL>
__global__ void __launch_bounds__(256)
k_bn_relu_planar(T* __restrict__ x, int C, int64_t inner, int chunks_per_plane, const float* __restrict__ mean,
                 const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, int relu)
{
    typedef T vec __attribute__((ext_vector_type(VL)));
    const int64_t plane = blockIdx.x / chunks_per_plane;
    const int chunk = blockIdx.x % chunks_per_plane;
    const int c = (int)(plane % C);
    const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    T* base = x + plane * inner;
    const int64_t n_vec = inner / VL;                        // (the host sends inner % VL != 0 through VL = 1)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int64_t i = ((int64_t)chunk * 4 + it) * 256 + threadIdx.x;
        if (i < n_vec) {
            vec* q = reinterpret_cast<vec*>(base + i * VL);
            vec v = *q;
#pragma unroll
            for (int j = 0; j < VL; ++j) v[j] = bn_relu_one<T>(v[j], m, is, ga, be, relu);
            *q = v;
        }
    }
}

extern "C" int sdirt_bn_relu(void* x, int64_t outer, int32_t channels, int64_t inner, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int32_t relu, int32_t half_precision, void* stream)
{
    if (!x || !mean || !invstd || !gamma || !beta || outer < 0 || channels < 1 || inner < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (outer * channels * inner == 0) return SDIRT_OK;
    const int vl = half_precision ? 8 : 4;
    const bool aligned = ((uintptr_t)x & 15) == 0;
    hipStream_t st = as_stream(stream);
    if (inner == 1) {
        // pixel-major: 16-byte groups of channels when the count allows it, one channel per thread otherwise
        const bool wide = aligned && channels % vl == 0 && channels / vl <= 256;
        if (!wide && channels > 256) return fail(SDIRT_ERR_UNSUPPORTED, "pixel-major tensors: channels <= 256 or a multiple of %d up to %d", vl, 256 * vl);
        const int groups = wide ? channels / vl : channels;
        const int bdim = groups * (256 / groups), ppb = bdim / groups;
        const int grid = (int)std::min<int64_t>((outer + ppb - 1) / ppb, 256 * 16);
#define SDIRT_BN(T, VL) k_bn_relu_pixel_major<T, VL><<<grid, bdim, 0, st>>>(static_cast<T*>(x), outer, channels, mean, invstd, gamma, beta, relu)
        if (half_precision) { if (wide) SDIRT_BN(_Float16, 8); else SDIRT_BN(_Float16, 1); }
        else { if (wide) SDIRT_BN(float, 4); else SDIRT_BN(float, 1); }
#undef SDIRT_BN
    } else {
        const bool wide = aligned && inner % vl == 0;
        const int64_t n_vec = wide ? inner / vl : inner;
        const int64_t cpp = (n_vec + 1023) / 1024, blocks = outer * channels * cpp;
        if (cpp > (1 << 30) || blocks > (1ll << 31) - 1) return fail(SDIRT_ERR_UNSUPPORTED, "tensor too large");
#define SDIRT_BN(T, VL) k_bn_relu_planar<T, VL><<<(unsigned)blocks, 256, 0, st>>>(static_cast<T*>(x), channels, inner, (int)cpp, mean, invstd, gamma, beta, relu)
        if (half_precision) { if (wide) SDIRT_BN(_Float16, 8); else SDIRT_BN(_Float16, 1); }
        else { if (wide) SDIRT_BN(float, 4); else SDIRT_BN(float, 1); }
#undef SDIRT_BN
    }
    LAUNCH_CHECK();
    return SDIRT_OK;
}

namespace {

// torch's linear-interpolation source index (UpSample.cuh: area_pixel_compute_source_index): align_corners -> scale * dst
// with scale = (in - 1) / (out - 1); else scale * (dst + 0.5) - 0.5 with scale = in / out, clamped at 0
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_at(int dst, int in, float scale, bool align)
{
    float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
    if (!align && src < 0.0f) src = 0.0f;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}
inline float lerp_scale(int in, int out, bool align)
{
    return align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f) : (float)in / (float)out;
}

constexpr int kMaxDispIn = 32, kMaxDispOut = 64;

// Disp + DisparityRegression (dddnet.py:543-568): trilinear interpolation of the hourglass's [B, 1, D0, H0, W0] volume to
// [D, H, W] (align_corners = False), softmin over the D shifts, expectation over shifts -D/2 .. D/2 - 1 -- one thread per
// output pixel, everything in fp32 (what autocast runs these ops in), ONE [B, 1, H, W] store instead of five passes over a
// [B, D, H, W] fp32 tensor.  Term order of the interpolation as torch's kernel (UpSampleTrilinear3d.cu).
// SD0 / SD > 0: the depth sizes as compile-time constants -- every loop unrolls, the depth weights and plane indices fold
// into the code and `plane` / `v` live in registers (the reference's 10 -> 20); 0: any sizes up to the bounds, arrays in scratch.
template <typename T, int SD0, int SD>
__global__ void __launch_bounds__(256)
k_disparity_regression(const T* __restrict__ cost, int B, int D0_, int H0, int W0, int D_, int H, int W, float sd_, float sh,
                       float sw, float* __restrict__ disp)
{
    constexpr bool FIXED = SD0 > 0;
    const int D0 = FIXED ? SD0 : D0_, D = FIXED ? SD : D_;
    const float sd = FIXED ? (float)SD0 / (float)(SD > 0 ? SD : 1) : sd_;
    const int64_t total = (int64_t)B * H * W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(t % W);
        const int h = (int)((t / W) % H);
        const int b = (int)(t / ((int64_t)W * H));
        const Lerp lh = lerp_at(h, H0, sh, false), lw = lerp_at(w, W0, sw, false);
        const T* base = cost + (int64_t)b * D0 * H0 * W0;
        float plane[FIXED ? SD0 : kMaxDispIn];           // the in-plane part of every input plane at (h, w)
#pragma unroll
        for (int k = 0; k < (FIXED ? SD0 : kMaxDispIn); ++k) {
            if (!FIXED && k >= D0) break;
            const T* p = base + (int64_t)k * H0 * W0;
            const float a = (float)p[lh.i0 * W0 + lw.i0], bb = (float)p[lh.i0 * W0 + lw.i1];
            const float c = (float)p[lh.i1 * W0 + lw.i0], d = (float)p[lh.i1 * W0 + lw.i1];
            plane[k] = lh.l0 * (lw.l0 * a + lw.l1 * bb) + lh.l1 * (lw.l0 * c + lw.l1 * d);
        }
        // softmin = softmax of the negated values: maximum first, as torch's softmax kernels
        float v[FIXED ? SD : kMaxDispOut];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < (FIXED ? SD : kMaxDispOut); ++i) {
            if (!FIXED && i >= D) break;
            const Lerp ld = lerp_at(i, D0, sd, false);
            v[i] = -(ld.l0 * plane[ld.i0] + ld.l1 * plane[ld.i1]);
            mx = fmaxf(mx, v[i]);
        }
        float sum = 0.0f, acc = 0.0f;
#pragma unroll
        for (int i = 0; i < (FIXED ? SD : kMaxDispOut); ++i) {
            if (!FIXED && i >= D) break;
            v[i] = expf(v[i] - mx);
            sum += v[i];
        }
        // torch.arange(-maxdisp // 2, maxdisp // 2): floor division of the NEGATED count (odd maxdisp: -(D+1)/2)
        const int first = -((D + 1) / 2);
#pragma unroll
        for (int i = 0; i < (FIXED ? SD : kMaxDispOut); ++i) {
            if (!FIXED && i >= D) break;
            acc += (v[i] / sum) * (float)(first + i);
        }
        disp[t] = acc;
    }
}

// nn.Upsample(scale_factor = 2, mode = 'trilinear', align_corners = True) of Conv2x (dddnet.py:585, 589) -- any output
// size -- on a volume stored [B, D, H, W, C] (channels_last_3d), 16 bytes of channels per thread: interpolated in fp32 from
// the tensor's values and rounded once to its type (under autocast torch runs the op in fp32 and the next convolution
// rounds its input to fp16: the same numbers).  torch's kernel is planar-only: in the channels_last_3d hourglass it stood
// between two layout copies.
template <typename T, int VL>
__global__ void __launch_bounds__(256)
k_upsample_trilinear_ndhwc(const T* __restrict__ x, int B, int C, int D0, int H0, int W0, int D, int H, int W, float sd, float sh,
                           float sw, T* __restrict__ out)
{
    typedef T vec __attribute__((ext_vector_type(VL)));
    const int groups = C / VL;
    const int64_t total = (int64_t)B * D * H * W * groups;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        int64_t rest = t / groups;
        const int w = (int)(rest % W); rest /= W;
        const int h = (int)(rest % H); rest /= H;
        const int d = (int)(rest % D);
        const int b = (int)(rest / D);
        const Lerp ld = lerp_at(d, D0, sd, true), lh = lerp_at(h, H0, sh, true), lw = lerp_at(w, W0, sw, true);
        const T* base = x + (int64_t)b * D0 * H0 * W0 * C + (int64_t)g * VL;
        auto at = [&](int dd, int hh, int ww) {
            return *reinterpret_cast<const vec*>(base + (((int64_t)dd * H0 + hh) * W0 + ww) * C);
        };
        const vec v000 = at(ld.i0, lh.i0, lw.i0), v001 = at(ld.i0, lh.i0, lw.i1), v010 = at(ld.i0, lh.i1, lw.i0),
                  v011 = at(ld.i0, lh.i1, lw.i1), v100 = at(ld.i1, lh.i0, lw.i0), v101 = at(ld.i1, lh.i0, lw.i1),
                  v110 = at(ld.i1, lh.i1, lw.i0), v111 = at(ld.i1, lh.i1, lw.i1);
        vec r;
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const float lo = lh.l0 * (lw.l0 * (float)v000[j] + lw.l1 * (float)v001[j]) +
                             lh.l1 * (lw.l0 * (float)v010[j] + lw.l1 * (float)v011[j]);
            const float hi = lh.l0 * (lw.l0 * (float)v100[j] + lw.l1 * (float)v101[j]) +
                             lh.l1 * (lw.l0 * (float)v110[j] + lw.l1 * (float)v111[j]);
            r[j] = (T)(ld.l0 * lo + ld.l1 * hi);
        }
        *reinterpret_cast<vec*>(out + t * VL) = r;
    }
}

}  // namespace

extern "C" int sdirt_disparity_regression(const void* cost, int32_t batch, int32_t d_in, int32_t h_in, int32_t w_in,
                                          int32_t d_out, int32_t h_out, int32_t w_out, int32_t half_precision, float* disp,
                                          void* stream)
{
    if (!cost || !disp || batch < 0 || d_in < 1 || h_in < 1 || w_in < 1 || d_out < 1 || h_out < 1 || w_out < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (d_in > kMaxDispIn || d_out > kMaxDispOut)
        return fail(SDIRT_ERR_UNSUPPORTED, "d_in <= %d and d_out <= %d (got %d, %d)", kMaxDispIn, kMaxDispOut, d_in, d_out);
    const int64_t total = (int64_t)batch * h_out * w_out;
    if (total == 0) return SDIRT_OK;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
    const float sd = lerp_scale(d_in, d_out, false), sh = lerp_scale(h_in, h_out, false), sw = lerp_scale(w_in, w_out, false);
    hipStream_t st = as_stream(stream);
#define SDIRT_DR(T, A, B_) k_disparity_regression<T, A, B_><<<grid, 256, 0, st>>>(static_cast<const T*>(cost), batch, d_in, h_in, w_in, \
                               d_out, h_out, w_out, sd, sh, sw, disp)
    // dddnet.py:112, 409-446: maxdisp 20, and the hourglass ends at the cost volume's own 20 planes (20 -> 20: the depth
    // weights fold to the identity, the interpolation is the in-plane x 4); 10 -> 20: the same network on a half-depth volume
    const int shape = d_out == 20 ? (d_in == 20 ? 2 : d_in == 10 ? 1 : 0) : 0;
    if (half_precision) { if (shape == 2) SDIRT_DR(_Float16, 20, 20); else if (shape == 1) SDIRT_DR(_Float16, 10, 20); else SDIRT_DR(_Float16, 0, 0); }
    else { if (shape == 2) SDIRT_DR(float, 20, 20); else if (shape == 1) SDIRT_DR(float, 10, 20); else SDIRT_DR(float, 0, 0); }
#undef SDIRT_DR
    LAUNCH_CHECK();
    return SDIRT_OK;
}

extern "C" int sdirt_upsample_trilinear_ndhwc(const void* x, int32_t batch, int32_t channels, int32_t d_in, int32_t h_in,
                                              int32_t w_in, int32_t d_out, int32_t h_out, int32_t w_out,
                                              int32_t half_precision, void* out, void* stream)
{
    if (!x || !out || batch < 0 || channels < 1 || d_in < 1 || h_in < 1 || w_in < 1 || d_out < 1 || h_out < 1 || w_out < 1)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (batch == 0) return SDIRT_OK;
    const int vl = half_precision ? 8 : 4;
    const bool wide = channels % vl == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    const int64_t total = (int64_t)batch * d_out * h_out * w_out * (channels / (wide ? vl : 1));
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
    const float sd = lerp_scale(d_in, d_out, true), sh = lerp_scale(h_in, h_out, true), sw = lerp_scale(w_in, w_out, true);
    hipStream_t st = as_stream(stream);
#define SDIRT_UP(T, VL) k_upsample_trilinear_ndhwc<T, VL><<<grid, 256, 0, st>>>(static_cast<const T*>(x), batch, channels, d_in, \
                            h_in, w_in, d_out, h_out, w_out, sd, sh, sw, static_cast<T*>(out))
    if (half_precision) { if (wide) SDIRT_UP(_Float16, 8); else SDIRT_UP(_Float16, 1); }
    else { if (wide) SDIRT_UP(float, 4); else SDIRT_UP(float, 1); }
#undef SDIRT_UP
    LAUNCH_CHECK();
    return SDIRT_OK;
}

// ---------------------------------------------------------------------------
// The context branches of the depth network's feature extractor (dfdp/dddnet/dddnet.py:376-385): nn.AvgPool2d with a
// window of 32 x 32 / 8 x 8 and the same stride on [B, 128, H/4, W/4] maps.  torch's kernel gives one THREAD the
// whole window (1024 strided reads): 240 us per call at 512 x 768, four calls per frame = 13 % of config 5's frame
// (profiles/r06/top_kernels_c5.txt).  Here one workgroup owns one output row of one plane: thread t sums column t
// of the k input rows (coalesced reads), the k column sums of a window are added from LDS.  fp32 accumulation and
// ONE division by k * k, as torch (accscalar_t = float); the order of the additions differs.
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_avg_pool_windows(const T* __restrict__ x, int H, int W, int k, T* __restrict__ out)
{
    extern __shared__ float colsum[];                           // [W]
    const int OH = H / k, OW = W / k;
    const int plane = blockIdx.x / OH, oy = blockIdx.x - plane * OH;
    const T* __restrict__ src = x + ((int64_t)plane * H + (int64_t)oy * k) * W;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int r = 0;
        for (; r + 3 < k; r += 4) {
            a0 += (float)src[(int64_t)r * W + c];
            a1 += (float)src[(int64_t)(r + 1) * W + c];
            a2 += (float)src[(int64_t)(r + 2) * W + c];
            a3 += (float)src[(int64_t)(r + 3) * W + c];
        }
        for (; r < k; ++r) a0 += (float)src[(int64_t)r * W + c];
        colsum[c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    const float inv = 1.0f / (float)(k * k);
    for (int ox = threadIdx.x; ox < OW; ox += blockDim.x) {
        float s = 0.0f;
        for (int j = 0; j < k; ++j) s += colsum[ox * k + j];
        out[((int64_t)plane * OH + oy) * OW + ox] = (T)(s * inv);
    }
}

// The same window average reading a pixel-major map [B, H, W, C] (channels_last) as it lies -> the planar [B, C, H/k, W/k] the
// other kernel writes (a few hundred values per plane: the 1 x 1 convolution behind it gets what it got before): one workgroup
// per output pixel, a thread owns a 16-byte group of channels and every (256 / groups)-th pixel of the window; the partial
// sums meet in LDS.  fp32 accumulation, one multiplication by 1 / k^2, like the planar kernel.
template <class T, int VL>
__global__ void __launch_bounds__(256) k_avg_pool_windows_nhwc(const T* __restrict__ x, int H, int W, int C, int k, T* __restrict__ out)
{
    typedef T vec __attribute__((ext_vector_type(VL)));
    extern __shared__ float part[];                             // [lanes][C]
    const int groups = C / VL, lanes = blockDim.x / groups;
    const int g = threadIdx.x % groups, lane = threadIdx.x / groups;
    const int OH = H / k, OW = W / k;
    const int ox = blockIdx.x % OW, oy = (blockIdx.x / OW) % OH, b = blockIdx.x / (OW * OH);
    const T* __restrict__ src = x + (((int64_t)b * H + (int64_t)oy * k) * W + (int64_t)ox * k) * C + g * VL;
    float acc[VL];
#pragma unroll
    for (int j = 0; j < VL; ++j) acc[j] = 0.0f;
#pragma unroll 4
    for (int p = lane; p < k * k; p += lanes) {
        const vec v = *reinterpret_cast<const vec*>(src + ((int64_t)(p / k) * W + p % k) * C);
#pragma unroll
        for (int j = 0; j < VL; ++j) acc[j] += (float)v[j];
    }
#pragma unroll
    for (int j = 0; j < VL; ++j) part[lane * C + g * VL + j] = acc[j];
    __syncthreads();
    const float inv = 1.0f / (float)(k * k);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s_ = 0.0f;
        for (int l = 0; l < lanes; ++l) s_ += part[l * C + c];
        out[(((int64_t)b * C + c) * OH + oy) * OW + ox] = (T)(s_ * inv);
    }
}

extern "C" int sdirt_avg_pool_windows_nhwc(const void* x, int32_t batch, int32_t height, int32_t width, int32_t channels, int32_t k,
                                           int32_t half_precision, void* out, void* stream)
{
    if (!x || !out || batch < 0 || height < 1 || width < 1 || channels < 1 || k < 1 || height % k || width % k)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument (the window must divide height and width)");
    const int vl = half_precision ? 8 : 4;
    if (channels % vl != 0 || channels / vl > 256 || ((uintptr_t)x & 15))
        return fail(SDIRT_ERR_UNSUPPORTED, "channels must be a multiple of %d (at most %d) and x 16-byte aligned", vl, 256 * vl);
    const int64_t blocks = (int64_t)batch * (height / k) * (width / k);
    if (blocks == 0) return SDIRT_OK;
    if (blocks > (1ll << 31) - 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "too many windows");
    const int groups = channels / vl, lanes = std::max(1, std::min(256 / groups, k * k));
    const size_t lds = sizeof(float) * (size_t)lanes * channels;
    if (half_precision)
        k_avg_pool_windows_nhwc<_Float16, 8><<<(unsigned)blocks, groups * lanes, lds, as_stream(stream)>>>(
            static_cast<const _Float16*>(x), height, width, channels, k, static_cast<_Float16*>(out));
    else
        k_avg_pool_windows_nhwc<float, 4><<<(unsigned)blocks, groups * lanes, lds, as_stream(stream)>>>(
            static_cast<const float*>(x), height, width, channels, k, static_cast<float*>(out));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

extern "C" int sdirt_avg_pool_windows(const void* x, int64_t planes, int32_t height, int32_t width, int32_t k,
                                      int32_t half_precision, void* out, void* stream)
{
    if (!x || !out || planes < 0 || height < 1 || width < 1 || k < 1 || height % k || width % k)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument (the window must divide height and width)");
    if (width > 12288) return fail(SDIRT_ERR_UNSUPPORTED, "width > 12288");
    const int64_t blocks = planes * (height / k);
    if (blocks == 0) return SDIRT_OK;
    if (blocks > (1ll << 31) - 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "too many planes");
    const size_t lds = sizeof(float) * (size_t)width;
    if (half_precision)
        k_avg_pool_windows<_Float16><<<(unsigned)blocks, 256, lds, as_stream(stream)>>>(
            static_cast<const _Float16*>(x), height, width, k, static_cast<_Float16*>(out));
    else
        k_avg_pool_windows<float><<<(unsigned)blocks, 256, lds, as_stream(stream)>>>(
            static_cast<const float*>(x), height, width, k, static_cast<float*>(out));
    LAUNCH_CHECK();
    return SDIRT_OK;
}

// ---------------------------------------------------------------------------
// PSFNet tone curves (deeplens/psfnet.py:589-620), one pass each instead of ~25 elementwise launches.
// The arithmetic follows torch's fp32 op sequence on the GPU term by term (scalar / tensor is
// reciprocal * scalar, tensor / scalar is tensor * (1 / scalar), no contraction), so the results
// are interchangeable with the op-by-op chain.
// ---------------------------------------------------------------------------
namespace {

constexpr float kA1 = 0.89129432f, kB1 = 0.27217316f, kC1 = -0.00246187f;           // psfnet.py:591
constexpr float kA2 = 5.94018909e-01f, kB2 = 1.20060450e+01f, kC2 = -5.24983855e-03f;  // psfnet.py:592

__device__ __forceinline__ float recip(float v) { return 1.0f / v; }

// degamma(img) = fit_degamma(img * 255): 8-bit code value -> luminance
__device__ __forceinline__ float tone_degamma(float img)
{
    const float x = img * 255.0f;
    const float lo = recip(recip(kA1 * x + kB1) + kC1);
    const float hi = recip(recip(kA2 * x + kB2) + kC2);
    const float t = fminf(x * (1.0f / 100.0f), 1.0f);
    return hi * t + lo * (1.0f - t);
}

// clip(gamma(l), 0, 1) with gamma(l) = fit_gamma(l) / 255
__device__ __forceinline__ float tone_gamma_clip(float l)
{
    const float r = recip(l + 1e-9f);
    const float x1 = (recip(r - kC1) - kB1) * (1.0f / kA1);
    const float x2 = (recip(r - kC2) - kB2) * (1.0f / kA2);
    const float t = fminf(((x1 + x2) * 0.5f) * (1.0f / 100.0f), 1.0f);
    const float g = (x2 * t + x1 * (1.0f - t)) * (1.0f / 255.0f);
    return fminf(fmaxf(g, 0.0f), 1.0f);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_tone(const float* __restrict__ in, int64_t n, float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = MODE == 0 ? tone_degamma(in[i]) : tone_gamma_clip(in[i]);
}

}  // namespace

extern "C" int sdirt_tone_curve(const float* in, int64_t n, int32_t mode, float* out, void* stream)
{
    if (!in || !out || n < 0 || mode < 0 || mode > 1) return fail(SDIRT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return SDIRT_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    if (mode == 0) k_tone<0><<<grid, 256, 0, as_stream(stream)>>>(in, n, out);
    else k_tone<1><<<grid, 256, 0, as_stream(stream)>>>(in, n, out);
    LAUNCH_CHECK();
    return SDIRT_OK;
}
