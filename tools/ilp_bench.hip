// ilp_bench.hip -- does instruction-level parallelism INSIDE a wave raise a gfx950 SIMD's issue rate?
// W waves on every SIMD (W workgroups of 256 threads per CU, limited by their LDS allocation) run a
// loop of 64 plain fp32 vector instructions arranged as ILP independent dependent chains
// (ILP = 1: one 64-long chain ... ILP = 4: four interleaved 16-long chains).  Reported: cycles per
// instruction per SIMD (wall clock x the shader clock the waves measured themselves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define I1(r) "v_mul_f32 " r ", " r ", %4\n\tv_add_f32 " r ", " r ", %5\n\tv_fma_f32 " r ", " r ", %4, %5\n\tv_mul_f32 " r ", " r ", %4\n\t"
// four instructions of each of two / four chains, interleaved instruction by instruction
#define I2 "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %5\n\t" \
           "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\t"
#define I4 "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t" \
           "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %5\n\tv_add_f32 %3, %3, %5\n\t" \
           "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t" \
           "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int ILP>
__global__ void __launch_bounds__(256) k_ilp(int reps, float* sink, unsigned long long* out)
{
    extern __shared__ float pad[];
    float a = 1.0f + threadIdx.x * 1e-3f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    const float c0 = 0.9999f, c1 = 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (ILP == 1) asm volatile(R16(I1("%0")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c0), "v"(c1));
        if (ILP == 2) asm volatile(R4(I2) R4(I2) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c0), "v"(c1));
        if (ILP == 4) asm volatile(R4(I4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c0), "v"(c1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (threadIdx.x == 1023) pad[0] = a;
    sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

int main()
{
    float* sink; unsigned long long* dcl;
    CHECK(hipMalloc(&sink, sizeof(float) * 2048 * 256)); CHECK(hipMalloc(&dcl, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 40000;
    printf("cycles per instruction per SIMD (64 plain fp32 vector instructions per loop trip)\n waves/SIMD   ILP 1    ILP 2    ILP 4\n");
    const int waves[] = {8, 6, 5, 4, 3, 2, 1};
    for (int W : waves) {
        const size_t lds = W == 8 ? 16 * 1024 : (size_t)(160 * 1024 / W) - 2048;     // W workgroups fit a CU's 160 KB, W + 1 do not
        printf("   %d      ", W);
        for (int ilp : {1, 2, 4}) {
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                auto k = ilp == 1 ? k_ilp<1> : ilp == 2 ? k_ilp<2> : k_ilp<4>;
                CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                CHECK(hipEventRecord(e0));
                k<<<256 * W, 256, lds>>>(reps, sink, dcl);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            unsigned long long h[2]; CHECK(hipMemcpy(h, dcl, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / h[1] / 10.0;
            printf("  %5.2f  ", ms * 1e6 * ghz / reps / 64.0 / W);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
