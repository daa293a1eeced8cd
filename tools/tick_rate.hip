// tick_rate.hip -- what does s_memtime count on gfx950?  One wave spins until N ticks of
// s_memtime (__builtin_readcyclecounter) have passed, also reading s_memrealtime (the constant
// 100 MHz counter), while the host times the launch: ticks per second, on an idle chip and with
// every SIMD busy with eight waves of dependent v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void k_spin(unsigned long long nticks, unsigned long long* out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < nticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

__global__ void __launch_bounds__(256) k_busy(int reps, float* sink, unsigned long long* out)
{
    float a = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                     : "+v"(a) : "v"(0.999f), "v"(1e-3f));
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = __builtin_readcyclecounter() - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    sink[blockIdx.x * 256 + threadIdx.x] = a;
}

int main()
{
    unsigned long long* d; float* sink;
    CHECK(hipMalloc(&d, 16)); CHECK(hipMalloc(&sink, sizeof(float) * 2048 * 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long h[2]; float ms;
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipEventRecord(e0)); k_spin<<<1, 64>>>(50000000ull, d); CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("idle chip, one wave: %llu s_memtime ticks, %llu s_memrealtime ticks in %.3f ms -> s_memtime %.1f MHz, s_memrealtime %.1f MHz\n",
               h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
    for (int W : {1, 4, 8}) {
        const int reps = 400000;
        CHECK(hipEventRecord(e0)); k_busy<<<256 * W, 256>>>(reps, sink, d); CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        const double per_instr_ns = ms * 1e6 / (8.0 * reps * W);           // per wave-instruction per SIMD
        printf("dependent v_fma_f32, %d waves/SIMD on every SIMD: wave 0 saw %llu s_memtime ticks / %llu realtime ticks in %.3f ms -> "
               "s_memtime %.1f MHz; %.3f ns per instruction per SIMD = %.1f TFLOP/s; %.2f s_memtime ticks per instruction per SIMD\n",
               W, h[0], h[1], ms, h[0] / ms / 1e3, per_instr_ns, 1024 * 128.0 / per_instr_ns / 1e3, (double)h[0] / (8.0 * reps * W));
    }
    return 0;
}
