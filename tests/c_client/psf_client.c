/* psf_client.c -- a C host program on the C ABI of libsdirt_dp.so (include/sdirt_dp.h), no Python anywhere:
 * Lensgroup.psf_diff(points, ks, spp, center=True) of deeplens/optics.py:934-996 as
 *
 *   sdirt_lens_create        the prescription at one wavelength (surface table + refractive indices)
 *   sdirt_points_to_object   normalised points -> object space (optics.py:959-960, 1302-1306)
 *   sdirt_psf_call           uniforms -> pupil points -> chief-ray pass + primary pass + splat + normalise, the
 *                            reference's batch-wide Newton trip rule evaluated on the device
 *   (status word != 0: call again with the tables the device derived -- SDIRT_CTL_TRIPS2 -- until it is 0)
 *
 * plain C99 + the HIP runtime for memory and the stream.  TEST INFRASTRUCTURE: tests/test_gpu_c_client.py writes the
 * input file from fixture F1 (the reference's own run of BASELINE config 1), builds this file with hipcc, runs it as
 * a child process and compares what it writes with the reference's PSF and with the oracle.
 *
 *   psf_client <input.bin> <output.bin>
 *
 * input.bin : struct client_header, then sdirt_surface_desc[n_surfaces], float points[n_points][3] (normalised:
 *             x, y in [-1, 1], z = depth in mm), float uniforms[2 spp + 2 spp_center] (the reference's draw order)
 * output.bin: int32 rounds, int32 trips[K], int32 trips_center[K], float center[n_points][2],
 *             float L[n_points][ks][ks], float R[n_points][ks][ks] (zeros when have_dp == 0)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdirt_dp.h"

struct client_header {
    int32_t magic, n_surfaces, n_points, spp, spp_center, ks, have_dp, flags;
    double pupil_r, pupil_r_center, pupil_z, d_sensor, pixel_size, tan_hfov, r_last, sensor_w, sensor_h;
    double dp[4];
};

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SD_OK(x) do { int r_ = (x); if (r_ != SDIRT_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, sdirt_last_error()); return 3; } } while (0)

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: psf_client <input.bin> <output.bin>\n"); return 1; }
    if (sdirt_abi_version() != SDIRT_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    struct client_header h;
    if (!f || fread(&h, sizeof h, 1, f) != 1 || h.magic != 0x53444952) { fprintf(stderr, "bad input\n"); return 1; }
    const int K = h.n_surfaces, N = h.n_points, S = h.spp, Sc = h.spp_center, ks = h.ks;
    const size_t n_u = 2 * (size_t)(S + Sc), tile = (size_t)ks * ks;
    sdirt_surface_desc* surf = malloc(sizeof *surf * K);
    float* pts = malloc(sizeof(float) * 3 * N);
    float* u_host = NULL;
    uint32_t* ctl_host = NULL;
    HIP_OK(hipHostMalloc((void**)&u_host, sizeof(float) * n_u, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&ctl_host, sizeof(uint32_t) * SDIRT_CTL_WORDS, hipHostMallocDefault));
    if (fread(surf, sizeof *surf, K, f) != (size_t)K || fread(pts, sizeof(float) * 3, N, f) != (size_t)N ||
        fread(u_host, sizeof(float), n_u, f) != n_u) { fprintf(stderr, "short input\n"); return 1; }
    fclose(f);

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    sdirt_lens* lens = NULL;
    SD_OK(sdirt_lens_create(surf, K, &lens));
    float *d_pts, *d_po, *d_cen, *d_l, *d_r;
    void* d_scratch;
    const int64_t scratch_bytes = sdirt_psf_call_scratch_bytes(N, S, Sc);
    HIP_OK(hipMalloc((void**)&d_pts, sizeof(float) * 3 * N));
    HIP_OK(hipMalloc((void**)&d_po, sizeof(float) * 3 * N));
    HIP_OK(hipMalloc((void**)&d_cen, sizeof(float) * 2 * N));
    HIP_OK(hipMalloc((void**)&d_l, sizeof(float) * tile * N));
    HIP_OK(hipMalloc((void**)&d_r, sizeof(float) * tile * N));
    HIP_OK(hipMalloc(&d_scratch, (size_t)scratch_bytes));
    HIP_OK(hipMemcpyAsync(d_pts, pts, sizeof(float) * 3 * N, hipMemcpyHostToDevice, st));
    SD_OK(sdirt_points_to_object(d_pts, N, h.tan_hfov, h.r_last, h.sensor_w, h.sensor_h, d_po, st));

    /* first bet: the loop's cap on every curved surface -- from above, one correction round is exact */
    int32_t trips[SDIRT_MAX_SURFACES], trips_c[SDIRT_MAX_SURFACES];
    for (int k = 0; k < K; ++k) trips[k] = trips_c[k] = surf[k].kind == SDIRT_PLANE ? 0 : SDIRT_NEWTON_MAXITER;
    sdirt_dp_params dp = {h.dp[0], h.dp[1], h.dp[2], h.dp[3]};
    int rounds = 0;
    for (;;) {
        ++rounds;
        SD_OK(sdirt_psf_call(lens, lens, d_po, N, u_host, S, Sc, h.pupil_r, h.pupil_r_center, h.pupil_z, h.d_sensor,
                             h.pixel_size, ks, h.have_dp ? &dp : NULL, trips, trips_c,
                             SDIRT_PSF_NORMALIZE | SDIRT_PSF_ZERO_CTL | (uint32_t)h.flags, d_cen, d_l, d_r, d_scratch,
                             ctl_host, st));
        HIP_OK(hipStreamSynchronize(st));
        if (ctl_host[SDIRT_CTL_ANY_VALID] != 1u) { fprintf(stderr, "No sampled rays is valid.\n"); return 4; }   /* optics.py:902 */
        if (ctl_host[SDIRT_CTL_STATUS] == 0u) break;
        if (rounds > 3 * K + 3) { fprintf(stderr, "trip tables did not settle\n"); return 5; }
        for (int k = 0; k < K; ++k) {
            trips[k] = (int8_t)(ctl_host[SDIRT_CTL_TRIPS2 + (k >> 2)] >> ((k & 3) * 8));
            trips_c[k] = (int8_t)(ctl_host[SDIRT_CTL_TRIPS2 + 16 + (k >> 2)] >> ((k & 3) * 8));
        }
    }
    float* cen = malloc(sizeof(float) * 2 * N);
    float* L = malloc(sizeof(float) * tile * N);
    float* R = malloc(sizeof(float) * tile * N);
    HIP_OK(hipMemcpy(cen, d_cen, sizeof(float) * 2 * N, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(L, d_l, sizeof(float) * tile * N, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(R, d_r, sizeof(float) * tile * N, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) return 1;
    int32_t r32 = rounds;
    fwrite(&r32, sizeof r32, 1, f);
    fwrite(trips, sizeof(int32_t), K, f);
    fwrite(trips_c, sizeof(int32_t), K, f);
    fwrite(cen, sizeof(float) * 2, N, f);
    fwrite(L, sizeof(float) * tile, N, f);
    fwrite(R, sizeof(float) * tile, N, f);
    fclose(f);
    printf("psf_client: %d point(s) x %d spp, ks %d, %d surfaces: %d round(s), status 0\n", N, S, ks, K, rounds);
    sdirt_lens_destroy(lens);
    return 0;
}
