// bw_read.hip -- what a read-only stream over 1.39 GB (the per-pixel PSF tensor of BASELINE config 5:
// 512 x 768 pixels x 2 x 21 x 21 fp32) reaches on this GPU: 16-byte loads, nothing else.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float fl4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const fl4* __restrict__ src, size_t n4, float* out)
{
    fl4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        fl4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * 256]) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = 1;
}

int main()
{
    const size_t bytes = 512ull * 768 * 2 * 441 * 4;
    fl4* d; float* o;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&o, 4)); CHECK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        for (int nt = 0; nt < 2; ++nt) {
            float best = 1e9f;
            for (int rep = 0; rep < 8; ++rep) {
                CHECK(hipEventRecord(e0));
                if (nt) k_read<8, true><<<grid, 256>>>(d, bytes / 16, o); else k_read<8, false><<<grid, 256>>>(d, bytes / 16, o);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("grid %6d x 256 threads, 8 x 16 B per thread in flight, %s loads: %.3f ms = %.0f GB/s\n", grid,
                   nt ? "non-temporal" : "plain", best, bytes / best / 1e6);
        }
    }
    return 0;
}
