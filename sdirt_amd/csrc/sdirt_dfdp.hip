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

}  // namespace

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
