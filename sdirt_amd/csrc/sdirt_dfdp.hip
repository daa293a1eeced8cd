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
