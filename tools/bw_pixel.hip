// bw_pixel.hip -- the READ pattern of k_local_psf_render_wave without its arithmetic: one wave per pixel, the pixel's
// [L | R] kernels (2 x 441 floats = 3528 bytes, 8-byte aligned) fetched one pixel ahead, as 14 dword loads per lane
// (the product's form), as 7 qword loads per lane (the 3528 bytes as 441 aligned 8-byte words) -- what would an 8-byte
// form of the renderer have to gain?  Same grid as the product: (W / 64, H) workgroups of 4 waves, 16 pixels per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/bw_pixel.hip -o /tmp/bw_pixel && /tmp/bw_pixel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float fl2 __attribute__((ext_vector_type(2)));
constexpr int KK = 441, H = 512, W = 768, CHUNK = 64, NWAVE = 4, PPW = CHUNK / NWAVE;

template <int WIDTH, int DEPTH>     // WIDTH 4: dword loads, 8: qword loads; DEPTH: pixels in flight ahead of the one consumed
__global__ void __launch_bounds__(256) k_pixels(const float* __restrict__ psf, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * CHUNK, row = blockIdx.y;
    const float* __restrict__ wrow = psf + (size_t)row * W * 2 * KK;
    constexpr int NI = WIDTH == 4 ? 14 : 7;
    float acc = 0.0f;
    float buf[DEPTH + 1][WIDTH == 4 ? 14 : 14];
    auto load = [&](int x, float (&b)[14]) {
        const float* __restrict__ k0 = wrow + (size_t)min(x, W - 1) * 2 * KK;
        if (WIDTH == 4) {
#pragma unroll
            for (int it = 0; it < 14; ++it) {
                const int f = min(it * 64 + lane, 2 * KK - 1);
                b[it] = __builtin_nontemporal_load(k0 + f);
            }
        } else {
            const fl2* __restrict__ q0 = reinterpret_cast<const fl2*>(k0);
#pragma unroll
            for (int it = 0; it < 7; ++it) {
                const int q = min(it * 64 + lane, KK - 1);
                const fl2 v = __builtin_nontemporal_load(q0 + q);
                b[2 * it] = v.x; b[2 * it + 1] = v.y;
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(x0 + wave + d * NWAVE, buf[d]);
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int x = x0 + wave + j * NWAVE;
        if (j + DEPTH < PPW) load(x + DEPTH * NWAVE, buf[(j + DEPTH) % (DEPTH + 1)]);
#pragma unroll
        for (int it = 0; it < 14; ++it) acc += buf[j % (DEPTH + 1)][it];
    }
    if (acc == 1234.5f) out[0] = acc;
    (void)NI;
}

int main()
{
    const size_t bytes = (size_t)H * W * 2 * KK * 4;
    float *d, *o;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&o, 4)); CHECK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(W / CHUNK, H);
    auto run = [&](const char* name, auto kern) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            CHECK(hipEventRecord(e0));
            kern<<<grid, 256>>>(d, o);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-44s %.4f ms = %.0f GB/s (%.3f of 8 TB/s)\n", name, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
    };
    run("dword loads (14 per lane), 1 pixel ahead", k_pixels<4, 1>);
    run("dword loads (14 per lane), 2 pixels ahead", k_pixels<4, 2>);
    run("qword loads (7 per lane), 1 pixel ahead", k_pixels<8, 1>);
    run("qword loads (7 per lane), 2 pixels ahead", k_pixels<8, 2>);
    run("qword loads (7 per lane), 3 pixels ahead", k_pixels<8, 3>);
    return 0;
}
