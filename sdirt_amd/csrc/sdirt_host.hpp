// Host-side helpers shared by the translation units of libsdirt_dp.so: the thread-local
// error text behind sdirt_last_error(), status macros, stream cast.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include <cstdint>
#include <vector>

#include "../../include/sdirt_dp.h"
#include "sdirt_device.hpp"

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(SDIRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

#define LAUNCH_CHECK()                                                                  \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess)                                                           \
            return fail(SDIRT_ERR_HIP, "kernel launch failed: %s (%s:%d)",              \
                        hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the CURRENT device (the binding makes the stream's device current for every call): 256 on an
// MI355X in SPX mode, 32 per XCD of a partitioned one.  Asked once per device, launch geometry is sized from it.
constexpr int kMaxDevices = 64;
inline int device_cus(int* out)
{
    static int cached[kMaxDevices] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < kMaxDevices && cached[dev] > 0) { *out = cached[dev]; return SDIRT_OK; }
    int n = 0;
    HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (n < 1) return fail(SDIRT_ERR_HIP, "device %d reports %d compute units", dev, n);
    if (dev >= 0 && dev < kMaxDevices) cached[dev] = n;
    *out = n;
    return SDIRT_OK;
}
// Opt a kernel in to more dynamic LDS than the default 48 KiB allowance (up to the CU's 160 KiB): once per kernel
// instantiation, device and size -- the largest size asked for so far is remembered per device, a later launch that
// needs more sets the attribute again.  (Concurrent host threads: the attribute call is idempotent, the remembered
// size only ever grows.)
template <auto Kernel>
inline int allow_large_lds(int bytes = 160 * 1024 - 1024)
{
    static std::atomic<int> allowed[kMaxDevices] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const bool known = dev >= 0 && dev < kMaxDevices;
    if (known && allowed[dev].load(std::memory_order_acquire) >= bytes) return SDIRT_OK;
    HIP_TRY(hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (known) {
        int seen = allowed[dev].load(std::memory_order_relaxed);
        while (seen < bytes && !allowed[dev].compare_exchange_weak(seen, bytes, std::memory_order_release)) {}
    }
    return SDIRT_OK;
}

// for the pure sizing helpers that may be asked before any device exists (sdirt_psf_spp_slices): MI355X's 256
inline int device_cus_or_default()
{
    int n = 0;
    return device_cus(&n) == SDIRT_OK ? n : 256;
}


// A prescription at one wavelength: the device table the kernels read and its host mirror.
struct sdirt_lens {
    int32_t n_surfaces;
    sdirt::DevSurface* dev;               // device table [n_surfaces]
    std::vector<sdirt::DevSurface> host;  // host mirror
};

// Newton trip counts of one launch, one signed byte per surface, passed by value at a FIXED
// offset of the kernel-argument segment; the kernels read the dword of surface k from there
// with a scalar load that shares the round trip of the surface's constant block (a byte-indexed
// by-value table made the compiler issue a vector load and wait for it once per surface).
struct alignas(64) TripTable {
    uint32_t w[SDIRT_MAX_SURFACES / 4];
};
static_assert(sizeof(TripTable) == 64, "layout");

inline int make_trips(const sdirt_lens* lens, const int32_t* trips, TripTable& tt)
{
    for (int k = 0; k < SDIRT_MAX_SURFACES / 4; ++k) tt.w[k] = 0;
    for (int k = 0; k < lens->n_surfaces; ++k) {
        int v = trips ? trips[k] : SDIRT_NEWTON_MAXITER;
        if (v < -SDIRT_NEWTON_MAXITER || v > SDIRT_NEWTON_MAXITER)
            return fail(SDIRT_ERR_INVALID_ARGUMENT, "trips[%d]=%d outside [-%d,%d]", k, v,
                        SDIRT_NEWTON_MAXITER, SDIRT_NEWTON_MAXITER);
        tt.w[k >> 2] |= (uint32_t)(uint8_t)(int8_t)v << ((k & 3) * 8);
    }
    return SDIRT_OK;
}

constexpr int kBlock = 256;
// Workgroup size of the two fused kernels.  512 threads = 8 waves share one pair of
// L/R tiles (33.8 KB at ks 65): 4 workgroups = 32 waves per CU = 8 per SIMD; both
// kernels need <= 43 VGPRs.  Measured on config 2: k_psf_lr 12.69 ms at 256 threads
// (LDS-limited to 4 waves/SIMD), 11.76 ms at 512, 15.7 ms at 1024.
constexpr int kFused = 512;

inline int grid_for(int64_t work, int block, int cap = 256 * 16)
{
    int64_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

inline int check_rays(const sdirt_rays& R)
{
    if (!R.ox || !R.oy || !R.oz || !R.dx || !R.dy || !R.dz || !R.ra)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "sdirt_rays has a null array");
    return SDIRT_OK;
}

// every component array on a 16-byte boundary (dwordx4 access)
inline bool rays_aligned16(const sdirt_rays& R)
{
    const uintptr_t a = (uintptr_t)R.ox | (uintptr_t)R.oy | (uintptr_t)R.oz | (uintptr_t)R.dx | (uintptr_t)R.dy |
                        (uintptr_t)R.dz | (uintptr_t)R.ra | (uintptr_t)R.obliq;
    return (a & 15) == 0;
}

inline int check_ks(int ks, int max_ks = SDIRT_MAX_KS)
{
    if (ks < 2 || ks > max_ks)
        return fail(SDIRT_ERR_INVALID_ARGUMENT, "ks=%d outside [2,%d]", ks, max_ks);
    return SDIRT_OK;
}
