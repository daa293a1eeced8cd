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
