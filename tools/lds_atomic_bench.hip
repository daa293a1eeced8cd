// lds_atomic_bench.hip -- what one LDS float atomic costs on gfx950 (tools; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o build/lds_atomic_bench && build/lds_atomic_bench
// Every wave issues ITER x 8 ds_add_f32 (or ds_add_u32 / ds_write_b32 / ds_add_rtn_f32) on addresses of a chosen
// pattern inside a 16900-dword tile; reported: LDS cycles per wave-instruction per CU (all CUs busy, W waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) k(const int* __restrict__ addr, int iters, float* __restrict__ out)
{
    extern __shared__ float tile[];
    for (int i = threadIdx.x; i < 16900; i += blockDim.x) tile[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int a[8];
    for (int j = 0; j < 8; ++j) a[j] = addr[j * 64 + lane];
    float v = 1.0f + lane * 1e-3f;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) atomicAdd(&tile[a[j]], v);
            else if (OP == 1) atomicAdd(reinterpret_cast<unsigned*>(tile) + a[j], (unsigned)lane);
            else if (OP == 2) reinterpret_cast<volatile float*>(tile)[a[j]] = v;
            else if (OP == 3) acc += atomicAdd(&tile[a[j]], v);
            else if (OP == 4) acc += reinterpret_cast<volatile float*>(tile)[a[j]];
            else if (OP == 5) atomicAdd(reinterpret_cast<unsigned long long*>(tile) + (a[j] >> 1), (unsigned long long)lane * 0x100000001ull);
            else if (OP == 6) atomicAdd(reinterpret_cast<double*>(tile) + (a[j] >> 1), (double)v);
        }
        v += 1e-6f;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[a[0]] + acc;
}

int main()
{
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("CUs %d, clock %.2f GHz (nominal)\n", ncu, clk_khz * 1e-6);
    const char* pat_name[] = {"lane i -> dword i (conflict-free)", "random in 64 dwords", "random in 128 dwords",
                              "random in 1024 dwords", "random in 16900 dwords", "all lanes one dword",
                              "16 lanes per dword (4 dwords)", "lane i -> dword 65*i (stride 65)",
                              "2 lanes per dword (32 distinct, distinct banks)"};
    const char* op_name[] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_add_rtn_f32", "ds_read_b32", "ds_add_u64", "ds_add_f64"};
    int* d_addr; float* d_out;
    CK(hipMalloc(&d_addr, 512 * sizeof(int)));
    CK(hipMalloc(&d_out, 4096 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int op = (getenv("OP0") ? atoi(getenv("OP0")) : 0); op < 7; ++op)
        for (int pat = 0; pat < 9; ++pat)
            for (int threads : {64, 256, 1024}) {
                std::vector<int> h(512);
                srand(1);
                for (int j = 0; j < 8; ++j)
                    for (int l = 0; l < 64; ++l) {
                        int v = 0;
                        switch (pat) {
                            case 0: v = l + 64 * j; break;
                            case 1: v = rand() % 64; break;
                            case 2: v = rand() % 128; break;
                            case 3: v = rand() % 1024; break;
                            case 4: v = rand() % 16900; break;
                            case 5: v = 7 + j; break;
                            case 6: v = (l / 16) * 4225 + j; break;
                            case 7: v = 65 * l + j; break;
                            case 8: v = (l / 2) + 64 * j; break;
                        }
                        h[j * 64 + l] = v;
                    }
                CK(hipMemcpy(d_addr, h.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
                auto launch = [&](int it) {
                    const size_t lds = 16900 * 4;
                    switch (op) {
                        case 0: k<0><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 1: k<1><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 2: k<2><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 3: k<3><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 4: k<4><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 5: k<5><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                        case 6: k<6><<<ncu, threads, lds>>>(d_addr, it, d_out); break;
                    }
                };
                launch(10);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                launch(iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double instr_per_cu = (double)iters * 8 * (threads / 64);
                printf("%-15s %-48s waves/CU %2d: %7.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", op_name[op],
                       pat_name[pat], threads / 64, ms * 1e-3 * 2.4e9 / instr_per_cu);
            }
    return 0;
}
