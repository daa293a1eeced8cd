// newton_bench.hip -- what does one Newton trip of the trace core cost on a gfx950 SIMD, and
// what does the SIMD actually issue per cycle?  ("attribute the idle VALU slots".)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//         tools/newton_bench.hip -o build/newton_bench && build/newton_bench
//
// Part 1 -- instruction microbenchmarks: R repetitions of a 32-instruction block of one kind,
// W single-wave workgroups per SIMD (grid = 1024 * W blocks of 64 threads), timed with s_memtime
// inside the kernel.  Reported: cycles per instruction seen by ONE wave, and the SIMD's throughput
// W / that.  "dep": one dependency chain; "ind": 8 independent chains.
// Part 2 -- the product's own newton_k<Lean, true, NoPoly> (included from sdirt_device.hpp, same
// compiler flags) run for many trips on register-resident rays: cycles per trip per wave and per
// SIMD at W = 1..8, next to the trip's static instruction mix.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../sdirt_amd/csrc/sdirt_device.hpp"

using namespace sdirt;

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

// One block of 32 (or 16 pairs of) instructions of one kind, repeated `reps` times.
// 256-thread workgroups: the four waves of a workgroup go to the four SIMDs of a CU, so a grid of
// 256 * W workgroups puts W waves on every SIMD.
template <int KIND>
__global__ void __launch_bounds__(256) k_inst(int reps, unsigned long long* out, float seed, float sc0)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
          a7 = a0 + 7;
    const float c0 = 0.999f, c1 = 1e-3f;
    double d0 = a0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) {          // v_fma_f32 dependent chain
            asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(c0), "v"(c1));
        } else if (KIND == 1) {   // v_fma_f32, 8 independent chains
            asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                              "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                              "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(c0), "v"(c1));
        } else if (KIND == 2) {   // v_mul_f32 (VOP2) dependent
            asm volatile(REP32("v_mul_f32 %0, %0, %1\n\t") : "+v"(a0) : "v"(c0));
        } else if (KIND == 3) {   // v_rcp_f32 dependent
            asm volatile(REP32("v_rcp_f32 %0, %0\n\t") : "+v"(a0));
        } else if (KIND == 4) {   // v_rcp_f32 independent
            asm volatile(REP4("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                              "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 5) {   // v_cmp (vcc) + v_cndmask, dependent pair x16
            asm volatile(REP4(REP4("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t"))
                         : "+v"(a0) : "v"(c0), "v"(c1) : "vcc");
        } else if (KIND == 6) {   // v_med3_f32 dependent
            asm volatile(REP32("v_med3_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(c1), "v"(c0));
        } else if (KIND == 7) {   // v_mul_f32 with an SGPR operand (VOP2), dependent
            asm volatile(REP32("v_mul_f32 %0, %1, %0\n\t") : "+v"(a0) : "s"(sc0));
        } else if (KIND == 8) {   // v_fma with an SGPR operand (VOP3), dependent
            asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "s"(sc0), "v"(c1));
        } else if (KIND == 9) {   // s_add_u32 chain (SALU), dependent
            unsigned m = (unsigned)r;
            asm volatile(REP32("s_add_u32 %0, %0, 1\n\t") : "+s"(m));
            a0 += (float)m;
        } else if (KIND == 10) {  // v_rsq_f32 dependent
            asm volatile(REP32("v_rsq_f32 %0, %0\n\t") : "+v"(a0));
        } else if (KIND == 11) {  // fma dependent chain interleaved with SALU (1:1)
            unsigned m = (unsigned)r;
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1\n\t"))
                         : "+v"(a0), "+s"(m) : "v"(c0), "v"(c1));
            a1 += (float)m;
        } else if (KIND == 12) {  // v_mul_f32 with an inline constant 0.5, dependent
            asm volatile(REP32("v_mul_f32 %0, 0.5, %0\n\t") : "+v"(a0));
        } else if (KIND == 13) {  // v_add_f32 with a 32-bit literal, dependent
            asm volatile(REP32("v_add_f32 %0, 0x3089705f, %0\n\t") : "+v"(a0));
        } else if (KIND == 14) {  // v_mul_f64 dependent
            asm volatile(REP32("v_mul_f64 %0, %0, %0\n\t") : "+v"(d0));
        } else if (KIND == 15) {  // v_cndmask with an SGPR-pair mask (VOP3), dependent
            asm volatile(REP32("v_cndmask_b32 %0, %0, %1, s[20:21]\n\t") : "+v"(a0) : "v"(c0) : "s20", "s21");
        } else if (KIND == 16) {  // v_cmp writing an SGPR pair (VOP3), independent of the chain
            asm volatile(REP32("v_cmp_lt_f32 s[20:21], %0, %1\n\t") : : "v"(a0), "v"(c0) : "s20", "s21");
        } else if (KIND == 17) {  // v_cmp to vcc (VOP2 encoding)
            asm volatile(REP32("v_cmp_lt_f32 vcc, %0, %1\n\t") : : "v"(a0), "v"(c0) : "vcc");
        } else if (KIND == 18) {  // two independent fma chains alternating (ILP 2)
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t"))
                         : "+v"(a0), "+v"(a1) : "v"(c0), "v"(c1));
        } else if (KIND == 19) {  // s_nop 0 x32
            asm volatile(REP32("s_nop 0\n\t"));
        } else if (KIND == 20) {  // v_fma_f32, three sources in three different VGPR banks (v41, v42, v43)
            asm volatile("v_mov_b32 v41, %0\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %2\n\t"
                         REP32("v_fma_f32 v41, v41, v42, v43\n\t") "v_mov_b32 %0, v41"
                         : "+v"(a0) : "v"(c0), "v"(c1) : "v41", "v42", "v43");
        } else if (KIND == 21) {  // v_fma_f32, three sources in the SAME bank (v40, v44, v48)
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v48, %2\n\t"
                         REP32("v_fma_f32 v40, v40, v44, v48\n\t") "v_mov_b32 %0, v40"
                         : "+v"(a0) : "v"(c0), "v"(c1) : "v40", "v44", "v48");
        } else if (KIND == 22) {  // v_mul_f32 (VOP2), two sources in the same bank (v40, v44)
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v44, %1\n\t"
                         REP32("v_mul_f32 v40, v40, v44\n\t") "v_mov_b32 %0, v40"
                         : "+v"(a0) : "v"(c0) : "v40", "v44");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)d0 == 1234.5f) out[0] = 0;
}

struct FakeSurf {
    u32x8 a;
    u32x4 b;
};

// newton_k on register-resident rays: `outer` calls of `trips` trips each
template <class M, int VARIANT>
__global__ void __launch_bounds__(256) k_newton(FakeSurf fs, int outer, int trips, unsigned long long* out,
                                               float* sink)
{
    Surf s;
    s.a = fs.a; s.b = fs.b;
    Ray r;
    const float u = (float)(threadIdx.x & 63) * (1.0f / 64.0f);
    r.ox = 3.0f * u - 1.5f; r.oy = 2.0f - 2.5f * u; r.oz = -1000.0f - 100.0f * u;
    r.dx = 0.002f * u; r.dy = -0.001f * u; r.dz = 1.0f;
    normalize3<M, true>(r.dx, r.dy, r.dz);
    r.ra = 1.0f; r.ob = 1.0f;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < outer; ++i) {
        float t;
        uint32_t mask;
        const bool v = newton_k<M, true>(s, NoPoly{}, r, trips, t, mask);
        acc += v ? t : 0.0f;
        r.ox += 1e-6f * t * 0.0f + 1e-7f;      // keep the loop from being hoisted
        asm volatile("" : "+v"(r.ox));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2)            // shader clock seen by one wave: ticks per 10 ns
        out[gridDim.x * 4] = (t1 - t0) * 100ull / (__builtin_amdgcn_s_memrealtime() - r0);
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

static uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static double median_cycles(std::vector<unsigned long long>& v)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

int main()
{
    unsigned long long* d_out;
    float* d_sink;
    const int maxb = 256 * 8;
    CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * (maxb * 4 + 1)));
    CHECK(hipMalloc(&d_sink, sizeof(float) * maxb * 256));
    std::vector<unsigned long long> h(maxb * 4);
    const char* names[] = {"v_fma_f32 dep", "v_fma_f32 ind8", "v_mul_f32 dep", "v_rcp_f32 dep", "v_rcp_f32 ind8",
                           "v_cmp vcc + s_nop 1 + v_cndmask (pair)", "v_med3_f32 dep", "v_mul_f32 sgpr operand dep",
                           "v_fma_f32 sgpr operand dep", "s_add_u32 dep (SALU)", "v_rsq_f32 dep",
                           "v_fma dep + s_add 1:1 (pair)", "v_mul_f32 inline const dep", "v_add_f32 literal dep",
                           "v_mul_f64 dep", "v_cndmask sgpr-pair mask dep", "v_cmp -> sgpr pair", "v_cmp -> vcc",
                           "v_fma_f32 two chains (ILP 2)", "s_nop 0", "v_fma_f32 srcs in 3 banks (v41,v42,v43)",
                           "v_fma_f32 srcs in 1 bank (v40,v44,v48)", "v_mul_f32 srcs in 1 bank (v40,v44)"};
    const int per_block[] = {32, 32, 32, 32, 32, 16, 32, 32, 32, 32, 32, 16, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};
    printf("# part 1: cycles per instruction as ONE wave sees them | cycles per instruction per SIMD (= that / W),\n"
           "#         W waves on every SIMD (256-thread workgroups, 256 * W of them)\n");
    printf("%-40s %11s %11s %11s %11s %11s\n", "kind", "W=1", "W=2", "W=4", "W=6", "W=8");
    const int reps = 2000;
    for (int kind = 0; kind < 23; ++kind) {
        printf("%-40s", names[kind]);
        for (int W : {1, 2, 4, 6, 8}) {
            const int grid = 256 * W;
#define LAUNCH(K) case K: k_inst<K><<<grid, 256>>>(reps, d_out, 1.0f, 0.999f); break;
            for (int pass = 0; pass < 2; ++pass) {
                switch (kind) {
                    LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
                    LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15) LAUNCH(16)
                    LAUNCH(17) LAUNCH(18) LAUNCH(19) LAUNCH(20) LAUNCH(21) LAUNCH(22)
                }
                CHECK(hipDeviceSynchronize());
            }
#undef LAUNCH
            h.resize(grid * 4);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
            const double cyc = median_cycles(h) / ((double)reps * per_block[kind]);
            printf(" %5.2f|%5.2f", cyc, cyc / W);
        }
        printf("\n");
    }

    // ---- part 2: the product's Newton loop
    FakeSurf fs;
    const float c = 1.0f / 35.0f, k = 0.0f, d = 0.0f, rlim = 14.0f;
    auto U = bits_of;
    fs.a = u32x8{1u | 8u | 16u | 4u, U(d), U(c), U(c * c), U(1.0f + k), U((1.0f / (c * c)) / (1.0f + k)),
                 U(rlim * rlim), U(rlim * rlim)};
    fs.b = u32x4{U(d + 1.0f / c), U(1.0f / 1.5f), U(1.0f / 2.25f), 0u};
    printf("\n# part 2: newton_k<M, k > -1, no polynomial>: cycles per trip (10-trip calls, the regain evaluation\n"
           "#         counted as an 11th trip): one wave's view | per SIMD\n");
    printf("%-28s %12s %12s %12s %12s %12s\n", "math", "W=1", "W=2", "W=4", "W=6", "W=8");
    float wall_ms[2][5];
    double clock_mhz[2][5];
    for (int variant = 0; variant < 2; ++variant) {
        int wi = 0;
        printf("%-28s", variant == 0 ? "Lean" : "Ieee");
        for (int W : {1, 2, 4, 6, 8}) {
            const int grid = 256 * W, outer = 200, trips = 10;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            float ms = 0.f;
            for (int pass = 0; pass < 2; ++pass) {
                CHECK(hipEventRecord(e0));
                if (variant == 0) k_newton<Lean, 0><<<grid, 256>>>(fs, outer, trips, d_out, d_sink);
                else k_newton<Ieee, 0><<<grid, 256>>>(fs, outer, trips, d_out, d_sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            wall_ms[variant][wi++] = ms;
            h.resize(grid * 4 + 1);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * (grid * 4 + 1), hipMemcpyDeviceToHost));
            clock_mhz[variant][wi - 1] = (double)h[grid * 4];
            h.resize(grid * 4);
            const double cyc = median_cycles(h) / ((double)outer * (trips + 1));
            printf(" %6.0f|%5.0f", cyc, cyc / W);
        }
        printf("\n");
    }
    printf("%-28s", "Lean: shader clock, MHz");
    for (int i = 0; i < 5; ++i) printf(" %12.0f", clock_mhz[0][i]);
    printf("   (s_memtime ticks per s_memrealtime tick x 100 MHz: the chip lowers its clock under this load)\n");
    // the same launches by the wall clock: kernel time / trips per wave = ns per trip per wave; x 2.4 GHz
    // = what the loop costs in nominal-clock cycles, i.e. in time (one trip at W = 8: this / 2.4 ns)
    for (int variant = 0; variant < 2; ++variant) {
        printf("%-28s", variant == 0 ? "Lean, wall clock x 2.4 GHz" : "Ieee, wall clock x 2.4 GHz");
        const int Ws[] = {1, 2, 4, 6, 8};
        for (int i = 0; i < 5; ++i) {
            const double cyc = wall_ms[variant][i] * 1e-3 * 2.4e9 / (200.0 * 11);
            printf(" %6.0f|%5.0f", cyc, cyc / Ws[i]);
        }
        printf("\n");
    }
    return 0;
}
