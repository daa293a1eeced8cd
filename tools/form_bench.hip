// form_bench.hip -- what one instruction FORM costs a gfx950 SIMD: W waves on every SIMD (W workgroups
// of 256 threads per CU, pinned by their LDS allocation, so that every SIMD holds exactly W waves) run
// a loop of 64 instructions of one form.  Reported: cycles per instruction per SIMD = wall time x the
// shader clock the waves measured themselves (s_memtime / s_memrealtime) / instructions per SIMD.
// (tools/newton_bench.hip timed a wave with s_memtime on a chip whose workgroups were not evenly
// spread; its per-form figures at 8 waves were too low.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

template <int KIND>
__global__ void __launch_bounds__(256) k_form(int reps, float sc, float* sink, unsigned long long* out)
{
    extern __shared__ float pad[];
    float a = 1.0f + threadIdx.x * 1e-3f, b = a + 0.5f;
    double da = a;
    const float c0 = 0.9999f, c1 = 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) asm volatile(R64("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 1) asm volatile(R64("v_mul_f32 %0, %1, %0\n\t") : "+v"(a) : "s"(sc));
        if (KIND == 2) asm volatile(R64("v_fma_f32 v8, v8, v4, v12\n\t") : : : "v4", "v8", "v12");
        if (KIND == 3) asm volatile(R64("v_cmp_lt_f32 vcc, %0, %1\n\t") : : "v"(a), "v"(c0) : "vcc");
        if (KIND == 4) asm volatile(R64("v_cndmask_b32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(c0), "s"(0x5555555555555555ull));
        if (KIND == 5) asm volatile(R64("v_med3_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 6) asm volatile(R64("v_mul_f64 %0, %0, %1\n\t") : "+v"(da) : "v"((double)c0));
        if (KIND == 7) asm volatile(R64("v_rcp_f32 %0, %0\n\t") : "+v"(a));
        if (KIND == 8) asm volatile(R16("v_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %2\n\tv_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %2\n\t") : "+v"(a), "+v"(b) : "v"(c0));
        if (KIND == 9) asm volatile(R64("s_add_u32 s40, s40, 1\n\t") : : : "s40", "scc");
        if (KIND == 10) asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s40, s40, 1\n\tv_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s40, s40, 1\n\t") : "+v"(a) : "v"(c0), "v"(c1) : "s40", "scc");
        if (KIND == 11) asm volatile(R16("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(c0), "v"(c1) : "vcc");
        if (KIND == 12) asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_rcp_f32 %0, %0\n\t") : "+v"(a) : "v"(c0), "v"(c1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (threadIdx.x == 1023) pad[0] = a;
    sink[blockIdx.x * 256 + threadIdx.x] = a + b + (float)da;
}

template <int KIND> static double run(int W, int reps, float* sink, unsigned long long* dcl, double* ghz_out)
{
    const size_t lds = W == 8 ? 16 * 1024 : (size_t)(160 * 1024 / W) - 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)k_form<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipEventRecord(e0));
        k_form<KIND><<<256 * W, 256, lds>>>(reps, 0.9999f, sink, dcl);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long h[2]; CHECK(hipMemcpy(h, dcl, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / h[1] / 10.0;
    *ghz_out = ghz;
    return ms * 1e6 * ghz / reps / 64.0 / W;
}

int main()
{
    float* sink; unsigned long long* dcl;
    CHECK(hipMalloc(&sink, sizeof(float) * 2048 * 256)); CHECK(hipMalloc(&dcl, 16));
    const char* names[] = {"v_fma_f32, VGPR operands, dependent chain", "v_mul_f32 with an SGPR operand", "v_fma_f32, three sources in one VGPR bank",
                           "v_cmp_lt_f32 -> vcc", "v_cndmask_b32, SGPR-pair mask", "v_med3_f32", "v_mul_f64", "v_rcp_f32, dependent",
                           "v_rcp_f32, independent", "s_add_u32 (scalar only)", "v_fma_f32 / s_add_u32 alternating (per instruction of either kind)",
                           "v_cmp, s_nop 1, v_cndmask(vcc), v_fma (per instruction, s_nop counted)", "3 v_fma_f32 + 1 v_rcp_f32, dependent"};
    printf("cycles per instruction per SIMD                                                    W=1     W=2     W=4     W=8   (clock at W=8)\n");
#define ROW(K) do { double g; printf("%-80s", names[K]); for (int W : {1, 2, 4, 8}) printf("  %6.2f", run<K>(W, 20000, sink, dcl, &g)); printf("   %.2f GHz\n", g); fflush(stdout); } while (0)
    ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(7); ROW(8); ROW(9); ROW(10); ROW(11); ROW(12);
    return 0;
}
