// mix_bench.hip -- does ONE expensive-form instruction slow its plain neighbours down?  Eight waves on
// every SIMD (256 * 8 workgroups of 256 threads) run a dependent chain of 64 vector instructions per
// loop trip: all plain (v_mul/v_add/v_fma on VGPRs), or the same with a few instructions replaced by
// a compare+select, an SGPR-operand multiply, a v_rcp_f32, or scalar instructions in between.
// Reported by the WALL clock: ns per loop trip per SIMD, and cycles at the shader clock the wave
// itself measured (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define P4 "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 %0, %0, %1\n\t"
#define P16 P4 P4 P4 P4
#define CMPSEL "v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t"
// the same chain with every operation spelled as v_fma_f32: a*b = fma(a, b, -0), a+b = fma(a, 1, b)
#define F4 "v_fma_f32 %0, %0, %1, %3\n\tv_fma_f32 %0, %0, 1.0, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %3\n\t"
#define F16 F4 F4 F4 F4
#define M4 "v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\t"
#define M16 M4 M4 M4 M4
#define A4 "v_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\t"
#define A16 A4 A4 A4 A4
#define MA4 "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t"
#define MA16 MA4 MA4 MA4 MA4
#define P14 P4 P4 P4 "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t"
#define P15 P4 P4 P4 "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"

template <int KIND>
__global__ void __launch_bounds__(256) k_mix(int reps, float sc, float* sink, unsigned long long* out)
{
    float a = 1.0f + threadIdx.x * 1e-3f;
    const float c0 = 0.9999f, c1 = 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) asm volatile(P16 P16 P16 P16 : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 1) asm volatile(P14 CMPSEL P14 CMPSEL P14 CMPSEL P14 CMPSEL : "+v"(a) : "v"(c0), "v"(c1) : "vcc");
        if (KIND == 2) asm volatile(P15 "v_mul_f32 %0, %3, %0\n\t" P15 "v_mul_f32 %0, %3, %0\n\t" P15 "v_mul_f32 %0, %3, %0\n\t" P15 "v_mul_f32 %0, %3, %0\n\t"
                                    : "+v"(a) : "v"(c0), "v"(c1), "s"(sc));
        if (KIND == 3) asm volatile(P15 "v_rcp_f32 %0, %0\n\t" P15 "v_rcp_f32 %0, %0\n\t" P15 "v_rcp_f32 %0, %0\n\t" P15 "v_rcp_f32 %0, %0\n\t"
                                    : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 4) asm volatile(P16 "s_add_u32 s40, s40, 1\n\t" P16 "s_add_u32 s40, s40, 1\n\t" P16 "s_add_u32 s40, s40, 1\n\t" P16 "s_add_u32 s40, s40, 1\n\t"
                                    : "+v"(a) : "v"(c0), "v"(c1) : "s40", "scc");
        if (KIND == 5) asm volatile(P4 P4 "v_rcp_f32 %0, %0\n\t" P4 CMPSEL "v_mul_f32 %0, %3, %0\n\t" P4 "s_add_u32 s40, s40, 1\n\t"
                                    P4 P4 "v_rcp_f32 %0, %0\n\t" P4 CMPSEL "v_mul_f32 %0, %3, %0\n\t" P4 "s_add_u32 s40, s40, 1\n\t"
                                    : "+v"(a) : "v"(c0), "v"(c1), "s"(sc) : "vcc", "s40", "scc");
        if (KIND == 6) asm volatile(F16 F16 F16 F16 : "+v"(a) : "v"(c0), "v"(c1), "v"(-0.0f));
        if (KIND == 7) asm volatile(M16 M16 M16 M16 : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 8) asm volatile(A16 A16 A16 A16 : "+v"(a) : "v"(c0), "v"(c1));
        if (KIND == 9) asm volatile(MA16 MA16 MA16 MA16 : "+v"(a) : "v"(c0), "v"(c1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[blockIdx.x * 256 + threadIdx.x] = a;
}

int main()
{
    float* sink; unsigned long long* d;
    CHECK(hipMalloc(&sink, sizeof(float) * 2048 * 256)); CHECK(hipMalloc(&d, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[] = {"64 plain (mul/add/fma chain)", "56 plain + 4 x (v_cmp, s_nop 1, v_cndmask)", "60 plain + 4 SGPR-operand v_mul",
                           "60 plain + 4 v_rcp_f32", "64 plain + 4 s_add_u32", "40 plain + 2 rcp + 2 cmp/sel + 2 sgpr-mul + 2 salu (a ray tracer's mix)",
                           "the 64-instruction chain of line 1 with every operation spelled v_fma_f32", "64 dependent v_mul_f32", "64 dependent v_add_f32",
                           "64 dependent, v_mul_f32 / v_add_f32 alternating"};
    const int ninstr[] = {64, 64, 64, 64, 68, 52, 64, 64, 64, 64};
    const int reps = 60000;
    for (int kind = 0; kind < 10; ++kind) {
        float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            CHECK(hipEventRecord(e0));
            switch (kind) {
            case 0: k_mix<0><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 1: k_mix<1><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 2: k_mix<2><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 3: k_mix<3><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 4: k_mix<4><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 5: k_mix<5><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 6: k_mix<6><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 7: k_mix<7><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 8: k_mix<8><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            case 9: k_mix<9><<<2048, 256>>>(reps, 0.9999f, sink, d); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long h[2]; CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        const double clock_ghz = (double)h[0] / h[1] / 10.0;
        const double ns_trip = ms * 1e6 / reps / 8.0;                 // per loop trip per SIMD (8 waves share it)
        printf("%-78s %7.1f ns/trip/SIMD  clock %.2f GHz  -> %6.1f cycles/trip/SIMD = %.2f per instruction\n", names[kind], ns_trip,
               clock_ghz, ns_trip * clock_ghz, ns_trip * clock_ghz / ninstr[kind]);
        fflush(stdout);
    }
    return 0;
}
