/*
 * sdirt_dp.h -- C ABI of libsdirt_dp.so, the MI355X (gfx950) dual-pixel
 * ray-traced PSF renderer.
 *
 * The reference (LinYark/Sdirt) has no native/FFI layer: its boundary for this
 * path is the Python call surface of deeplens/optics.py, surfaces.py and
 * monte_carlo.py.  Each entry point below replaces one of those calls; the
 * reference file:line it stands in for is cited on the declaration.  The
 * Python package sdirt_amd/ binds these symbols with ctypes and re-exposes the
 * reference's signatures (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - plain C types only; every pointer marked `dev` is a DEVICE pointer
 *    (hipMalloc'd or a torch.cuda tensor's data_ptr()); `host` pointers are
 *    read during the call and may be freed right after it returns;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - data-path calls never allocate, never free and never synchronise: they
 *    enqueue kernels on `stream` and return.  Only sdirt_lens_create /
 *    sdirt_lens_destroy touch the allocator;
 *  - every function returns SDIRT_OK (0) or a negative sdirt_status; the text
 *    of the last error on the calling thread is returned by sdirt_last_error();
 *  - rays are stored SoA (sdirt_rays): 8 arrays of S*N floats, element
 *    [s*N + n] = sample s of point source n -- consecutive lanes read
 *    consecutive addresses.  The reference keeps o,d as [S,N,3]
 *    (deeplens/basics.py:216-245); the shim converts at the API edge only.
 */
#ifndef SDIRT_DP_H
#define SDIRT_DP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDIRT_ABI_VERSION 4
#define SDIRT_MAX_SURFACES 64
#define SDIRT_MAX_AI 8
#define SDIRT_NEWTON_MAXITER 10 /* deeplens/surfaces.py:26 */
#define SDIRT_MAX_WAVELENGTHS 3 /* wavelength slots of one fused launch (psf_rgb) */
#define SDIRT_MAX_KS 141        /* two ks*ks fp32 tiles + 1 KiB of bookkeeping fit in 160 KiB of LDS */
#define SDIRT_MAX_KS_STAGED 1024 /* sdirt_forward_integral adds into the grids in HBM once its tiles no longer fit LDS: no LDS bound
                                   (the reference's draw_mtf asks for ks 256, optics.py:2056) */

typedef enum sdirt_status {
    SDIRT_OK = 0,
    SDIRT_ERR_INVALID_ARGUMENT = -1,
    SDIRT_ERR_UNSUPPORTED = -2, /* e.g. conic k <= -1 together with a feature not built */
    SDIRT_ERR_HIP = -3,         /* a HIP runtime call failed, see sdirt_last_error() */
    SDIRT_ERR_NO_DEVICE = -4
} sdirt_status;

typedef enum sdirt_surface_kind {
    SDIRT_PLANE = 0,  /* c == 0: stop / flat        deeplens/surfaces.py:409 */
    SDIRT_SPHERE = 1, /* ai is None and k == 0      deeplens/surfaces.py:456 */
    SDIRT_ASPHERE = 2 /* everything else            deeplens/surfaces.py:491 */
} sdirt_surface_kind;

/* One optical surface, as deeplens/surfaces.py:291-331 (Aspheric.__init__) holds
 * it: r is a Python float (double), d/c/k/ai are fp32 tensors.  n1/n2 are
 * Material.ior(wvln) of the medium before/after the surface, float64
 * (deeplens/basics.py:316-340). */
typedef struct sdirt_surface_desc {
    int32_t kind;      /* sdirt_surface_kind */
    int32_t ai_degree; /* 0, or number of even-asphere terms ai2, ai4, ... (<= 8) */
    double r;          /* semi-aperture [mm] */
    float d;           /* vertex z [mm] */
    float c;           /* curvature 1/roc [1/mm] */
    float k;           /* conic constant */
    float ai[SDIRT_MAX_AI];
    double n1;
    double n2;
} sdirt_surface_desc;

/* A lens prescription at ONE wavelength, resident on the device. */
typedef struct sdirt_lens sdirt_lens;

/* SoA ray bundle: the reference's Ray(o, d, ra, obliq) (deeplens/basics.py:216-245).
 *
 * Order in memory.  A bundle of M rays is eight arrays of M floats.  For the [spp, n_points] bundles of the PSF path
 * (sdirt_sample_rays, sdirt_center_from_rays, sdirt_forward_integral; n_points > 1 in sdirt_rays_from_aos / _to_aos)
 * the order is POINT-MAJOR: ray (s, n) -- sample s of point n, the reference's tensor element [s, n] -- is element
 * n * spp + s: the rays of one point are contiguous.  Sampling, tracing and propagation are per-ray and read / write
 * every array front to back whatever the order; the per-point reductions (centroid, splat) then read each point's
 * rays as one contiguous run, which the reference's own [spp, n_points] order would scatter at a stride of
 * n_points floats (measured: 8 to 32 useful bytes per 128-byte line and workgroup, profiles/r04).  The Python
 * binding presents the reference's [spp, n_points(, 3)] views on top (sdirt_amd/basics.py: Ray). */
typedef struct sdirt_rays {
    float* ox; float* oy; float* oz; /* dev, M each */
    float* dx; float* dy; float* dz; /* dev, M each, unit length */
    float* ra;                       /* dev, weight (validity 0/1 on traced rays) */
    float* obliq;                    /* dev, product of cos(refraction angles); may be NULL */
} sdirt_rays;

/* Dual-pixel sensor model parameters: the reference's param_list
 * [h, f, w, r, direct] without the direction letter
 * (deeplens/monte_carlo.py:157-164).  NULL where a `const sdirt_dp_params*` is
 * taken means param_list=None: defaults h=0.78 f=1.44 w=0.3 r=0.5 and, as in
 * the reference (monte_carlo.py:231), the R grid is left all-zero. */
typedef struct sdirt_dp_params {
    double h, f, w, r;
} sdirt_dp_params;

/* ---- flags ---------------------------------------------------------------- */
#define SDIRT_PSF_NORMALIZE 1u /* apply optics.py:983-987 to each written grid */
/* By default the fused kernels divide and take square roots with "lean" sequences (rcp /
 * sqrt seed + fma corrections, 6 and 10 instructions) that are PROVEN bit-identical to
 * correctly rounded IEEE results for normal-range operands: all 2^46 mantissa pairs for the
 * division, every fp32 >= 2^-100 for the square root (sdirt_selftest_math, tools/
 * selftest_math.py, profiles/r01/selftest_math.txt).  They skip the range scaling and
 * special-value fix-up of the compiler's 12/17-instruction sequences: x/0 yields NaN
 * instead of inf, denormal operands are not handled.  No valid ray produces such operands.
 * This flag selects the compiler's full-range IEEE sequences instead (~1.26x slower). */
#define SDIRT_PSF_STRICT_IEEE 4u
/* sdirt_trace only: read every surface's constants with a plain load-and-wait at the top of the
 * surface instead of the hand-scheduled prefetch one surface ahead (sdirt_device.hpp: surf_issue /
 * surf_wait).  Same arithmetic, hence bit-identical rays: the binding traces a probe bundle both
 * ways when it first uploads a prescription and refuses to go on if they differ (a miscompiled
 * prefetch would trace with stale constants -- the build-time ISA check is the first guard,
 * this is the second). */
#define SDIRT_TRACE_NO_PREFETCH 8u
/* sdirt_psf_lr_verified / sdirt_psf_call only: verify on the device (status word, corrected tables) but do NOT
 * enqueue round 2 -- for callers whose speculated tables have been right for a long streak: three empty
 * launches less per call; when the status comes back non-zero the caller launches the correction itself. */
#define SDIRT_PSF_ONE_ROUND 16u
/* sdirt_psf_lr / _centered / _verified / sdirt_psf_call: l_psf and r_psf are the two halves of ONE [N, 2, ks, ks] array --
 * point n's left grid at l_psf + n * 2 * ks * ks, its right grid ks * ks floats behind it; r_psf must be l_psf + ks * ks,
 * dp != NULL, one wavelength.  A rank of a multi-GPU volume renders its shard straight into the block that ONE all-gather
 * moves (`[N/8, 2, ks, ks]`), and L = a[:, 0], R = a[:, 1] are views of the gathered array: no staging copy. */
#define SDIRT_PSF_INTERLEAVED 32u
/* sdirt_psf_call only: the call clears the control block itself (inside its pupil-mapping launch) -- for a caller that
 * keeps ONE scratch block per in-flight step and reuses it step after step. */
#define SDIRT_PSF_ZERO_CTL 64u
/* sdirt_psf_call only (one workgroup per point): leave the trip rule unevaluated -- the masks of a rank of a sharded
 * batch say nothing before they are OR-ed with the other ranks' (sdirt_ctl_to_lanes, all-reduce, sdirt_ctl_from_lanes). */
#define SDIRT_PSF_NO_VERIFY 128u
/* sdirt_psf_lr / _centered / sdirt_psf_call (one workgroup per point): run-to-run IDENTICAL grids also on 50 to 70 pixels
 * (L + R; L alone: up to 99) -- BASELINE config 2's 65 x 65.  A point's grids are summed in LDS: in float64 (rounded to
 * fp32 once: the arrival order of the atomics cannot matter) wherever four workgroups per CU have room for such tiles,
 * ks <= 49; above, the default is fp32 tiles, whose sums differ in the last bits from run to run.  With this flag the
 * tiles stay float64 and the workgroups get 1024 threads, two per CU: +0.5 % time (profiles/r05/ab_tiles.txt).  The
 * chief-ray centroid is then reduced over 1024 instead of 512 partial sums (float64; the fp32 centre it rounds to may
 * differ from the default path's in the last bit).  SDIRT_ERR_UNSUPPORTED beyond ks 70 / 99, for r > 0.5 and when the
 * spp axis is cut (sdirt_psf_spp_slices > 1: partial grids meet in global float atomics). */
#define SDIRT_PSF_DETERMINISTIC 256u

/* ---- library ------------------------------------------------------------ */
int sdirt_abi_version(void);
const char* sdirt_last_error(void);
/* Number of visible HIP devices (0 if none); does not initialise a context. */
int sdirt_device_count(void);

/* ---- lens ---------------------------------------------------------------- */
/* Replaces Lensgroup.read_lens_json + per-call Material.ior evaluation
 * (deeplens/optics.py:2173-2198, deeplens/surfaces.py:399-405): builds the flat
 * per-surface constant block (fp32 thresholds rounded exactly where torch
 * rounds them) and uploads it to the current device. */
int sdirt_lens_create(const sdirt_surface_desc* surfaces /*host*/, int32_t n_surfaces,
                      sdirt_lens** out_lens);
void sdirt_lens_destroy(sdirt_lens* lens);
int32_t sdirt_lens_num_surfaces(const sdirt_lens* lens);

/* ---- staged path (same decomposition as the reference) ------------------ */

/* Lensgroup.psf_diff, deeplens/optics.py:956-960 + calc_scale_pinhole :1302-1306:
 * normalised points [N,3] (x,y in [-1,1], z = depth mm) -> object-space points.
 * sensor_w/h = sensor_size[1], sensor_size[0]. */
int sdirt_points_to_object(const float* points /*dev [N,3]*/, int64_t n_points, double tan_hfov,
                           double r_last, double sensor_w, double sensor_h,
                           float* point_obj /*dev [N,3]*/, void* stream);

/* Lensgroup.sample_from_points, deeplens/optics.py:482-488: uniforms in [0,1)
 * -> points on the pupil disc of radius pupil_r. */
int sdirt_pupil_samples(const float* u_theta /*dev [S]*/, const float* u_r2 /*dev [S]*/,
                        int64_t spp, double pupil_r, float* x2 /*dev [S]*/, float* y2 /*dev [S]*/,
                        void* stream);

/* Lensgroup.sample_from_points + Ray.__init__, deeplens/optics.py:479,490-494,
 * deeplens/basics.py:238-245: rays from every point to every pupil sample,
 * normalised; ra = obliq = 1.  Point-major: ray (s, n) is element n * spp + s. */
int sdirt_sample_rays(const float* point_obj /*dev [N,3]*/, int64_t n_points,
                      const float* x2 /*dev [S]*/, const float* y2 /*dev [S]*/, int64_t spp,
                      double pupil_z, sdirt_rays rays, void* stream);

/* Ray.__init__, deeplens/basics.py:233-245: build a ray bundle from the
 * reference's AoS tensors o,d [M,3] (d is L2-normalised with eps 1e-12 when
 * normalize != 0); ra (dev [M]) may be NULL (= all ones); obliq is set to 1.
 * n_points <= 1: ray j of the tensors is element j of the bundle.  n_points > 1: the tensors are the reference's
 * [spp, n_points, 3] (spp = n_rays / n_points; ra [spp, n_points]) and the bundle is point-major. */
int sdirt_rays_from_aos(const float* o /*dev [M,3]*/, const float* d /*dev [M,3]*/,
                        const float* ra /*dev [M] or NULL*/, int64_t n_rays, int64_t n_points, int32_t normalize,
                        sdirt_rays rays, void* stream);

/* Inverse view for callers that read ray.o / ray.d as [..., 3] tensors (n_points as above). */
int sdirt_rays_to_aos(sdirt_rays rays, int64_t n_rays, int64_t n_points, float* o /*dev [M,3] or NULL*/,
                      float* d /*dev [M,3] or NULL*/, void* stream);

/* Lensgroup.trace / _forward_tracing / _backward_tracing,
 * deeplens/optics.py:601-627,666-717, with Aspheric.ray_reaction
 * (deeplens/surfaces.py:391-520) per surface: traces surfaces [first,last) in
 * increasing order (backward == 0) or decreasing order (backward != 0), in place.
 *
 * trips (host, one int per lens surface, indexed by surface): number of Newton
 * loop trips to run on that surface.  The reference runs the loop until EVERY
 * ray of the batch has |f(t)| <= 50e-6 or 10 trips (surfaces.py:547), so the
 * count is a property of the whole batch; conv_mask (dev, one uint32 per lens
 * surface, zeroed by the caller, may be NULL) receives, OR-ed over all rays,
 * bit j (1..trips) = "some ray still had |f(t)| > 50e-6 in trip j".  A trip
 * table T reproduces the reference's batch exactly iff for every surface bits
 * 1..T-1 are set and bit T is clear (or T == 10); sdirt_amd/newton.py runs that
 * check and re-launches with the corrected table when speculation fails.
 * A NEGATIVE entry -T selects the speed mode for that surface: at most T trips, and every 64-ray
 * wave leaves the loop as soon as none of ITS rays is open -- the reference's loop condition per
 * wave instead of per batch; no host check is needed, results differ from the batch-exact ones
 * in the low-order bits of t (PSFs to ~1e-5 of their peak). */
int sdirt_trace(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                const int32_t* trips /*host [K]*/,
                uint32_t flags /*SDIRT_PSF_STRICT_IEEE | SDIRT_TRACE_NO_PREFETCH or 0*/, sdirt_rays rays, int64_t n_rays, uint32_t* conv_mask /*dev [K] or NULL*/,
                void* stream);

/* The same trace OUT OF PLACE: reads `rays`, writes `out` (another bundle of n_rays rays; `out` == `rays` is
 * sdirt_trace).  The reference's trace mutates its Ray; a caller that may have to run a batch again with a corrected
 * trip table keeps its input this way without copying it first (Lensgroup.trace: the copy was 64 bytes of traffic per
 * ray on top of the trace's own 64). */
int sdirt_trace_to(const sdirt_lens* lens, int32_t first, int32_t last, int32_t backward,
                   const int32_t* trips /*host [K]*/, uint32_t flags, sdirt_rays rays, sdirt_rays out,
                   int64_t n_rays, uint32_t* conv_mask /*dev [K] or NULL*/, void* stream);

/* Lensgroup.trace2sensor, deeplens/optics.py:638-664: the forward trace through ALL surfaces and the propagation to the
 * sensor plane z = d_sensor in one pass over the bundle (reads `rays`, writes `out`; the same bundle for in place).
 * Equal, bit for bit, to sdirt_trace_to(0, K, forward) followed by sdirt_propagate_to(d_sensor) under the default math
 * policy (the propagation's one division is the policy's: correctly rounded either way). */
int sdirt_trace2sensor(const sdirt_lens* lens, const int32_t* trips /*host [K]*/, uint32_t flags /*SDIRT_PSF_STRICT_IEEE or 0*/,
                       double d_sensor, sdirt_rays rays, sdirt_rays out, int64_t n_rays,
                       uint32_t* conv_mask /*dev [K] or NULL*/, void* stream);

/* Ray.propagate_to, deeplens/basics.py:256-264. */
int sdirt_propagate_to(double z, sdirt_rays rays, int64_t n_rays, void* stream);

/* Centroid part of Lensgroup.psf_center, deeplens/optics.py:902-904:
 * center[n] = -(sum_s o*ra / (sum_s ra + 1e-9)).xy (sums in float64);  any_valid (dev int32,
 * zeroed by caller, may be NULL) is set to 1 if any ray has ra == 1.  Point-major bundle. */
int sdirt_center_from_rays(sdirt_rays rays, int64_t spp, int64_t n_points,
                           float* center /*dev [N,2]*/, int32_t* any_valid /*dev*/, void* stream);

/* forward_integral + assign_points_to_pixels_small_r / _big_r,
 * deeplens/monte_carlo.py:9-68, 135-240, 242-372: sensor-plane rays -> RAW left
 * and right grids [N,ks,ks] (fully overwritten).  r_grid may be NULL.  Point-major bundle (ray (s, n) = element
 * n * spp + s).  ps = pixel size; center = pointc_ref [N,2].  ks up to SDIRT_MAX_KS_STAGED.  Where a point's grids are
 * summed is decided by what fits 159 KiB of LDS (sdirt_forward_integral_plan says which): float64 accumulators whenever
 * ntile * ks * ks of them fit (ntile = 2 with an R grid, else 1: L + R up to ks 100, L alone up to ks 142), summed in
 * arrival order and rounded to fp32 once; else float accumulators (L + R up to ks 142 -- SDIRT_MAX_KS and one more --, L
 * alone up to ks 201); else the contributions are added to the grids in HBM with float atomics.  (The fused
 * sdirt_psf_* entries stop at SDIRT_MAX_KS; the staged chain sample -> trace -> chief centre -> forward_integral ->
 * normalize is the path for larger grids.)  SDIRT_PSF_NORMALIZE: the grids leave max-normalised
 * (deeplens/optics.py:983-987, what sdirt_psf_normalize does) -- straight from the tiles in LDS when a workgroup holds a
 * point's whole sum: the 2 x 4 bytes per pixel of the separate pass are never read back. */
int sdirt_forward_integral(sdirt_rays rays, int64_t spp, int64_t n_points, double ps, int32_t ks,
                           const float* center /*dev [N,2]*/, const sdirt_dp_params* dp /*host*/,
                           uint32_t flags /*SDIRT_PSF_STRICT_IEEE | SDIRT_PSF_NORMALIZE or 0*/,
                           float* l_grid /*dev [N,ks,ks]*/, float* r_grid /*dev or NULL*/,
                           void* stream);

/* How sdirt_forward_integral launches on a device of n_cus compute units (host arithmetic only, for callers that size
 * scratch or want to know the path taken, and for the CPU tests): plan[0] = bytes per accumulator -- 8: float64 tiles in
 * LDS, 4: float tiles in LDS, 0: grids added to in HBM --, plan[1] = points per workgroup, plan[2] = workgroups along the
 * points, plan[3] = slices of the spp axis (> 1: partial tiles are added to the zeroed output), plan[4] = samples per
 * slice, plan[5] = LDS bytes per workgroup.  both != 0: L and R grids. */
int sdirt_forward_integral_plan(int64_t n_points, int64_t spp, int32_t ks, int32_t both, int32_t n_cus,
                                int64_t* plan /*host [6]*/);

/* deeplens/optics.py:983-987: psf / (max + 1e-6), per point, in place. */
int sdirt_psf_normalize(float* psf /*dev [N,ks,ks]*/, int64_t n_points, int32_t ks, void* stream);

/* ---- fused hot path ------------------------------------------------------ */

/* Chief-ray centre, Lensgroup.psf_center(method='chief_ray'),
 * deeplens/optics.py:898-904, fused: sample (xc,yc are pupil samples on the
 * shrunk pupil) -> trace to sensor -> centroid, rays never leave registers. */
int sdirt_chief_center(const sdirt_lens* lens, const float* point_obj /*dev [N,3]*/,
                       int64_t n_points, const float* xc /*dev [Sc]*/, const float* yc /*dev [Sc]*/,
                       int64_t spp_center, double pupil_z, double d_sensor,
                       const int32_t* trips /*host [K]*/, uint32_t flags /*SDIRT_PSF_STRICT_IEEE or 0*/,
                       float* center /*dev [N,2]*/, int32_t* any_valid /*dev or NULL*/,
                       uint32_t* conv_mask /*dev [K] or NULL*/, void* stream);



/* Lensgroup.psf_diff, deeplens/optics.py:934-996, fused from sampling to the
 * normalised left/right PSFs: one workgroup per (point, spp-slice), rays
 * generated in registers, per-surface constants read through the scalar cache,
 * L/R tiles accumulated with LDS float atomics, one coalesced store.
 * center = pointc_ref [N,2] (from sdirt_chief_center, or the pinhole centre
 * for center=False, optics.py:973-976).  r_psf may be NULL. */
int sdirt_psf_lr(const sdirt_lens* lens, const float* point_obj /*dev [N,3]*/, int64_t n_points,
                 const float* x2 /*dev [S]*/, const float* y2 /*dev [S]*/, int64_t spp,
                 double pupil_z, double d_sensor, double ps, int32_t ks,
                 const float* center /*dev [N,2]*/, const sdirt_dp_params* dp /*host or NULL*/,
                 const int32_t* trips /*host [K]*/, uint32_t flags, float* l_psf /*dev [N,ks,ks]*/,
                 float* r_psf /*dev or NULL*/, uint32_t* conv_mask /*dev [K] or NULL*/,
                 void* stream);

/* Lensgroup.psf_diff with center=True in ONE call: the chief-ray pass (sdirt_chief_center,
 * through lens_center = the lens at the default wavelength, optics.py:900) and the primary
 * pass.  When one workgroup owns a point (always, unless few points carry very many samples)
 * both passes run inside a single kernel launch -- no second launch, no host round trip between
 * them; otherwise the centre kernel is enqueued first.  center [N,2] receives the centres;
 * conv_mask / conv_mask_center as in sdirt_trace, one per pass. */
int sdirt_psf_lr_centered(const sdirt_lens* lens, const sdirt_lens* lens_center,
                          const float* point_obj /*dev [N,3]*/, int64_t n_points,
                          const float* x2 /*dev [S]*/, const float* y2 /*dev [S]*/, int64_t spp,
                          const float* xc /*dev [Sc]*/, const float* yc /*dev [Sc]*/,
                          int64_t spp_center, double pupil_z, double d_sensor, double ps, int32_t ks,
                          const sdirt_dp_params* dp /*host or NULL*/,
                          const int32_t* trips /*host [K]*/, const int32_t* trips_center /*host [K]*/,
                          uint32_t flags, float* center /*dev [N,2], out*/,
                          int32_t* any_valid /*dev or NULL*/, float* l_psf /*dev [N,ks,ks]*/,
                          float* r_psf /*dev or NULL*/, uint32_t* conv_mask /*dev [K] or NULL*/,
                          uint32_t* conv_mask_center /*dev [K] or NULL*/, void* stream);

/* Lensgroup.psf_rgb, deeplens/optics.py:999-1015 (and psf_map, :1018-1041, on top of it), as ONE
 * kernel launch: n_wvln (<= SDIRT_MAX_WAVELENGTHS) independent psf_diff calls -- one lens table,
 * one primary and one chief-ray pupil sample set (the reference draws fresh samples per
 * wavelength), one Newton trip table and one convergence-mask row per wavelength slot
 * (gridDim.y = n_wvln), every chief-ray pass through lens_center (optics.py:900).
 * Layouts: lens host [n_wvln]; x2 / y2 dev [n_wvln][S]; xc / yc dev [n_wvln][Sc]; trips /
 * trips_center host [n_wvln][K]; center dev [n_wvln][N][2]; any_valid dev [n_wvln];
 * l_psf / r_psf dev [N][n_wvln][ks][ks] (the reference's torch.stack(..., dim=-3));
 * conv_mask / conv_mask_center dev [n_wvln][SDIRT_MAX_SURFACES].
 * One workgroup per (point, wavelength) whatever N and S are, so the chief-ray pass is always fused. */
int sdirt_psf_rgb_centered(const sdirt_lens* const* lens /*host [n_wvln]*/, int32_t n_wvln,
                           const sdirt_lens* lens_center, const float* point_obj /*dev [N,3]*/,
                           int64_t n_points, const float* x2, const float* y2, int64_t spp,
                           const float* xc, const float* yc, int64_t spp_center, double pupil_z,
                           double d_sensor, double ps, int32_t ks, const sdirt_dp_params* dp,
                           const int32_t* trips, const int32_t* trips_center, uint32_t flags,
                           float* center, int32_t* any_valid, float* l_psf, float* r_psf,
                           uint32_t* conv_mask, uint32_t* conv_mask_center, void* stream);

/* Lensgroup.psf_rgb(center=False), deeplens/optics.py:999-1015 with :972-976, as ONE kernel launch:
 * sdirt_psf_lr for n_wvln wavelength slots (gridDim.y = n_wvln, one workgroup per (point, wavelength)),
 * the PSFs centred on caller-supplied centres (the pinhole image points) instead of chief rays.
 * Layouts as sdirt_psf_rgb_centered: lens host [n_wvln]; x2 / y2 dev [n_wvln][S]; trips host [n_wvln][K];
 * center dev [n_wvln][N][2]; l_psf / r_psf dev [N][n_wvln][ks][ks]; conv_mask dev [n_wvln][SDIRT_MAX_SURFACES]. */
int sdirt_psf_rgb(const sdirt_lens* const* lens /*host [n_wvln]*/, int32_t n_wvln,
                  const float* point_obj /*dev [N,3]*/, int64_t n_points, const float* x2, const float* y2,
                  int64_t spp, double pupil_z, double d_sensor, double ps, int32_t ks,
                  const float* center /*dev [n_wvln][N][2]*/, const sdirt_dp_params* dp /*host or NULL*/,
                  const int32_t* trips /*host [n_wvln][K]*/, uint32_t flags, float* l_psf, float* r_psf /*dev or NULL*/,
                  uint32_t* conv_mask /*dev or NULL*/, void* stream);

/* ---- speculate, verify on the device, re-render once ---------------------- */

/* How sdirt_psf_lr_centered cuts the spp axis for (n_points, spp): 1 = one workgroup per point (the
 * chief-ray pass runs inside the same kernel); > 1 = few points with many samples (the PSFNet
 * fitting loop: 64 points x 20000 spp, deeplens/psfnet.py:101-167), several workgroups per point.
 * n_cus > 0: pure host arithmetic for a device with that many compute units (MI355X: 256; a CPU-only caller);
 * n_cus <= 0: asks the CURRENT device (initialises the HIP runtime) -- what the launching entries use.  The slice
 * count sets the order in which partial grids are added (global float atomics), so the last bits of a split call's
 * PSFs depend on the CU count of the device that rendered them. */
int32_t sdirt_psf_spp_slices(int64_t n_points, int64_t spp, int32_t n_cus);

/* Control block of sdirt_psf_lr_verified: SDIRT_CTL_WORDS uint32 words at the start of `scratch`. */
#define SDIRT_CTL_STATUS 0     /* 0: the speculated tables were the reference's (round 2 did nothing);       */
                               /* else bit 0 | bit 1 (primary table corrected) | bit 2 (chief-ray table corrected) */
#define SDIRT_CTL_ANY_VALID 1  /* 1 if any chief ray reached the sensor (optics.py:902)                      */
#define SDIRT_CTL_TAG 2        /* sdirt_ctl_from_lanes: the caller's 30-bit tag of sdirt_ctl_to_lanes (e.g. a checksum of the   */
                               /* step's uniforms) and, in word 3, its complement to 0x3fffffff, as the MAX over ranks left     */
                               /* them: the two still add up to 0x3fffffff iff every rank handed in the same tag               */
#define SDIRT_CTL_TRIPS2 16    /* 16 + 16 words: the tables round 2 ran, one signed byte per surface         */
#define SDIRT_CTL_MASKS 64     /* 4 x 64 words: convergence masks of round 1 (primary, chief), round 2 (same) */
#define SDIRT_CTL_WORDS 320
int64_t sdirt_psf_verified_scratch_bytes(int64_t n_points, int64_t spp_center);

/* sdirt_psf_lr_centered for the case sdirt_psf_spp_slices(n_points, spp) > 1, WITHOUT a host round trip
 * when the speculated trip tables turn out wrong: round 1 renders with `trips` / `trips_center`; its
 * last kernel checks both tables against the convergence masks the round produced (the rule of
 * sdirt_trace, evaluated on the device) and writes the corrected tables into the control block;
 * round 2 -- the same kernels, enqueued right behind -- re-renders with them, or returns at once
 * when round 1 was right.  The host reads the control block once: status, the tables that were
 * run, the masks of both rounds.  (A table speculated from above is corrected exactly in one round;
 * if round 2's masks still disagree the caller goes on as with sdirt_psf_lr_centered.)
 * The chief-ray pass runs in slices too (partial sums in `scratch`, added in slice order).
 * scratch: dev, 8-byte aligned, sdirt_psf_verified_scratch_bytes(n_points, spp_center) bytes, its first
 * SDIRT_CTL_WORDS words zeroed by the caller.  Returns SDIRT_ERR_UNSUPPORTED when
 * sdirt_psf_spp_slices(n_points, spp) == 1. */
int sdirt_psf_lr_verified(const sdirt_lens* lens, const sdirt_lens* lens_center,
                          const float* point_obj /*dev [N,3]*/, int64_t n_points,
                          const float* x2 /*dev [S]*/, const float* y2 /*dev [S]*/, int64_t spp,
                          const float* xc /*dev [Sc]*/, const float* yc /*dev [Sc]*/, int64_t spp_center,
                          double pupil_z, double d_sensor, double ps, int32_t ks,
                          const sdirt_dp_params* dp /*host or NULL*/, const int32_t* trips /*host [K]*/,
                          const int32_t* trips_center /*host [K]*/, uint32_t flags,
                          float* center /*dev [N,2], out*/, float* l_psf /*dev [N,ks,ks]*/,
                          float* r_psf /*dev or NULL*/, void* scratch /*dev*/, void* stream);

/* One whole psf call enqueued by ONE library call (a Python caller pays per call into the library; a C caller needs
 * nothing else): upload the 2 * spp + 2 * spp_center uniforms the caller drew (u_host, page-locked, in the reference's
 * draw order -- theta[spp], r2[spp] of Lensgroup.sample_from_points, optics.py:483-484, then theta[spp_center],
 * r2[spp_center] of the same lines inside psf_center, optics.py:898), map them onto the pupil and the shrunk pupil, render
 * with the speculated tables, evaluate the reference's batch-wide trip rule ON THE DEVICE, and copy the control block to
 * ctl_host (page-locked, SDIRT_CTL_WORDS words; NULL: no copy).  The caller synchronises the stream and reads ctl_host:
 *   - few points with many samples (sdirt_psf_spp_slices > 1): sdirt_psf_lr_verified -- a wrong table has already been
 *     corrected and re-rendered behind round 1 (unless SDIRT_PSF_ONE_ROUND);
 *   - one workgroup per point (sdirt_psf_spp_slices == 1; any batch from one point to a whole volume): ONE fused launch;
 *     SDIRT_CTL_STATUS != 0 means the tables were not the reference's for this batch: call again with the tables in
 *     SDIRT_CTL_TRIPS2 (one signed byte per surface: primary at word 16, chief-ray at word 32) until the status is 0 --
 *     a table speculated from above (10 trips on curved surfaces, what NULL-free first calls should pass) is corrected
 *     exactly in one round.  No host-side rule is needed: tests/c_client/psf_client.c renders fixture F1 this way.
 * u_host is read by the call's first kernel WHERE IT IS (page-locked memory is mapped into the device's address space: no copy
 * command on the stream, whose two engine hand-overs cost a 2048-point step 25 us): it must stay unchanged until that kernel has
 * run -- one buffer per call in flight.  (Memory that is not mapped is copied instead.)
 * scratch: dev, 8-byte aligned, sdirt_psf_call_scratch_bytes(n_points, spp, spp_center) bytes: [control block | chief-ray
 * partial sums | uniforms | pupil points x2, y2, xc, yc]; its first SDIRT_CTL_WORDS words zeroed by the caller, or by
 * the call itself under SDIRT_PSF_ZERO_CTL. */
int64_t sdirt_psf_call_scratch_bytes(int64_t n_points, int64_t spp, int64_t spp_center);
int sdirt_psf_call(const sdirt_lens* lens, const sdirt_lens* lens_center, const float* point_obj /*dev [N,3]*/,
                   int64_t n_points, const float* u_host /*host, page-locked*/, int64_t spp, int64_t spp_center,
                   double pupil_r, double pupil_r_center, double pupil_z, double d_sensor, double ps, int32_t ks,
                   const sdirt_dp_params* dp /*host or NULL*/, const int32_t* trips /*host [K]*/,
                   const int32_t* trips_center /*host [K]*/, uint32_t flags, float* center /*dev [N,2], out*/,
                   float* l_psf /*dev [N,ks,ks]*/, float* r_psf /*dev or NULL*/, void* scratch /*dev*/,
                   uint32_t* ctl_host /*host, page-locked, out, or NULL*/, void* stream);

/* A batch sharded over ranks: the reference's trip rule is batch-wide, so the ranks' masks are OR-ed
 * before it is evaluated.  RCCL has no bitwise OR: sdirt_ctl_to_lanes spreads round 1's masks and the any-valid flag of
 * a control block (sdirt_psf_call with SDIRT_PSF_NO_VERIFY) into SDIRT_CTL_LANES int32 lanes -- 0 / 1 lanes [primary |
 * chief-ray][SDIRT_MAX_SURFACES][bit 0..10], the flag, then `tag` (30 bits: whatever the ranks must agree on, e.g. a checksum of the
 * uniforms they drew) and its complement --, the caller all-reduces them with MAX, and
 * sdirt_ctl_from_lanes folds them back into the control block, evaluates the rule there (lens != NULL: status word and
 * corrected tables as sdirt_psf_call leaves them; trips / trips_center = the tables that ran) and copies the block to
 * ctl_host (page-locked, or NULL).  Every rank then reads the same status.  Device pointers + stream, no allocation. */
#define SDIRT_CTL_LANES 1411   /* 2 * SDIRT_MAX_SURFACES * (SDIRT_NEWTON_MAXITER + 1) + 3 */
int sdirt_ctl_to_lanes(const uint32_t* ctl /*dev*/, uint32_t tag, int32_t* lanes /*dev [SDIRT_CTL_LANES], out*/, void* stream);
int sdirt_ctl_from_lanes(const int32_t* lanes /*dev*/, const sdirt_lens* lens /*or NULL: masks only*/,
                         const int32_t* trips /*host [K]*/, const int32_t* trips_center /*host [K]*/,
                         uint32_t* ctl /*dev, in/out*/, uint32_t* ctl_host /*host, page-locked, out, or NULL*/, void* stream);

/* ---- host side: the reference's random stream ------------------------------ */

/* torch.rand(n) on the CPU default generator -- the draws of Lensgroup.sample_from_points,
 * deeplens/optics.py:483-484 -- without torch's per-number loop: th_state is the byte image
 * torch.get_rng_state() returns (a copy the caller hands back with torch.set_rng_state); on return
 * out[0..n) holds the n floats torch.rand(n) would have produced from that state and th_state the state
 * torch would be in afterwards (MT19937 blocks of 624, (x & 0xffffff) * 2^-24 per number).  Host memory
 * only, no GPU involved.  SDIRT_ERR_UNSUPPORTED when the state image does not look as expected. */
int sdirt_host_uniform_fill(void* th_state /*host, in/out*/, int64_t state_bytes, int64_t n, float* out /*host*/);

/* ---- diagnostics ----------------------------------------------------------- */

/* Counts how often the lean arithmetic (the default, see SDIRT_PSF_STRICT_IEEE) differs from correctly rounded IEEE:
 * mode 0 = sqrt on every fp32 bit pattern in [first, first+count) (exhaustive for 0, 2^32);
 * mode 1 = division on `count` pseudo-random operand pairs (all mantissas, exponents within
 * +-exp_span of 0); mode 2 = division on mantissa pairs [first, first+count) of all 2^46.
 * out (dev, 9 x uint64): out[0] = mismatches, out[1..8] = examples. */
int sdirt_selftest_math(int32_t mode, uint64_t first, uint64_t count, int32_t exp_span,
                        uint64_t* out /*dev [9]*/, void* stream);

/* ---- image-space consumer of the PSFs ------------------------------------ */

/* local_psf_render_fast / local_psf_render / local_dp_psf_render,
 * deeplens/render_psf.py:76-188: per-pixel L/R PSF convolution with replicate
 * padding and flipped kernels.  img [B,C,H,W] fp32, psf [B,H,W,2,ks,ks] fp32;
 * half_precision != 0 reproduces the fp16 arithmetic of the _fast variant
 * (inputs rounded to fp16, products rounded to fp16, fp32 accumulation, result
 * rounded to fp16), 0 the fp32 arithmetic of local_dp_psf_render. */
int sdirt_local_psf_render(const float* img /*dev*/, const float* psf /*dev*/, int32_t batch,
                           int32_t channels, int32_t height, int32_t width, int32_t ks,
                           int32_t half_precision, float* out_l /*dev [B,C,H,W]*/,
                           float* out_r /*dev [B,C,H,W]*/, void* stream);

/* PSFNet.pred (deeplens/psfnet.py:317-336: stack(net(x,y,z), fliplr(net(-x,y,z))), each side
 * divided by its own sum + 1e-9) followed by local_psf_render_fast (render_psf.py:120-155), as
 * PSFNet.render chains them (psfnet.py:702-707), in one pass over the network's raw outputs.
 * raw_l, raw_r: fp16 [B*H*W, ks*ks], 16-byte aligned; the flipped / normalised per-pixel kernels
 * are formed in LDS and never written to memory.  Same fp16 arithmetic as
 * sdirt_local_psf_render(half_precision=1).  A kernel whose raw sum is 0 renders 0. */
int sdirt_psfnet_render(const float* img /*dev [B,C,H,W]*/, const void* raw_l /*dev fp16*/,
                        const void* raw_r /*dev fp16*/, int32_t batch, int32_t channels,
                        int32_t height, int32_t width, int32_t ks, float* out_l /*dev [B,C,H,W]*/,
                        float* out_r /*dev [B,C,H,W]*/, void* stream);

/* ---- the PSF network itself ------------------------------------------------ */

/* The network is described by its layer widths: widths[0..n_layers] = in, hidden..., out.  The
 * kernel is built for the reference's MLP (deeplens/psfnet_arch.py:26-50; psfnet.py:78):
 * 3 -> h4 -> 512 -> ... -> 512 -> out with h4 in {32, 64, 96, 128}, n_layers >= 3, out <= 512
 * (the reference: 3 -> 128 -> 512 x 9 -> ks*ks); other shapes return SDIRT_ERR_UNSUPPORTED. */

/* Bytes of the packed form of the whole network (weights as fp16 MFMA fragments, then biases
 * as fp32), or -1 if the shape is unsupported. */
int64_t sdirt_mlp_packed_bytes(const int32_t* widths /*host [n_layers+1]*/, int32_t n_layers);

/* nn.Linear weights (fp32 [out, in] row-major) and biases (fp32 [out]) of every layer ->
 * `packed`.  Per layer: out padded to 512 rows (128 for the first layer), in to a multiple of 16
 * columns, zero filled; tile (mt, ks) of 32 outputs x 16 inputs is one contiguous 1 KB block, lane
 * l = 32 h + r holding W[32 mt + r][16 ks + 8 h + j], j < 8.  Call again whenever the weights change. */
int sdirt_mlp_pack(const float* const* weights /*host array of dev ptrs*/,
                   const float* const* biases /*host array of dev ptrs*/,
                   const int32_t* widths /*host*/, int32_t n_layers,
                   void* packed /*dev, 16-byte aligned, out*/, void* stream);

/* MLP.forward under fp16 autocast (psfnet_arch.py:46-50: fp16 operands, fp32 accumulation,
 * bias, ReLU after EVERY layer, activations rounded to fp16) for all rows in one kernel, the
 * activations of 128 rows at a time held in LDS from the first layer to the last.
 * inp: dev fp32 [n_points, 3].  mirror != 0 appends the rows (-x, y, z) after the n_points rows
 * (x, y, z) -- the two passes of PSFNet.pred, psfnet.py:327-329.
 * out: dev fp16 [(1 + mirror) * n_points, out_features], 16-byte aligned. */
int sdirt_psfnet_mlp(const void* packed /*dev*/, const int32_t* widths /*host*/, int32_t n_layers,
                     const float* inp /*dev*/, int64_t n_points, int32_t mirror,
                     void* out /*dev fp16*/, void* stream);

/* ---- depth-from-dual-pixel network: cost volume ---------------------------- */

/* YRStereonet_3D.forward's cost volume (dfdp/dddnet/dddnet.py:136-148): x, y [B,C,H,W] left / right
 * feature maps -> cost [B,2C,D,H,W]; plane i holds, for gap = i - D/2, x in channels [0,C) and y
 * displaced by gap columns in channels [C,2C), both zero in the |gap| columns the shift vacates
 * (columns >= W+gap for gap < 0, columns < gap for gap > 0).  half_precision != 0: fp16 tensors
 * (the network runs under autocast), else fp32.  Every output element is written once. */
int sdirt_dp_cost_volume(const void* x /*dev*/, const void* y /*dev*/, int32_t batch,
                         int32_t channels, int32_t d_max, int32_t height, int32_t width,
                         int32_t half_precision, void* cost /*dev, out*/, void* stream);

/* The same volume between pixel-major tensors: x, y stored [B,H,W,C] (what a channels_last [B,C,H,W] tensor is in memory)
 * -> cost stored [B,D,H,W,2C] (a channels_last_3d [B,2C,D,H,W] tensor), element for element what sdirt_dp_cost_volume
 * writes (dfdp/dddnet/dddnet.py:136-148).  The layout the hourglass's 3-D convolutions (dddnet.py:409-446) compute in:
 * nothing is transposed between the feature network and the first of them.  16-byte accesses when `channels` is a
 * multiple of 8 (fp16) / 4 (fp32) and the three pointers are 16-byte aligned, element-wise otherwise. */
int sdirt_dp_cost_volume_nhwc(const void* x /*dev*/, const void* y /*dev*/, int32_t batch,
                              int32_t channels, int32_t d_max, int32_t height, int32_t width,
                              int32_t half_precision, void* cost /*dev, out*/, void* stream);

/* Adjoint of sdirt_dp_cost_volume for training the depth network: grad_cost [B,2C,D,H,W] ->
 * grad_x, grad_y [B,C,H,W] (each input element sums the gradients of the volume elements it was
 * copied to, fp32 accumulation). */
int sdirt_dp_cost_volume_backward(const void* grad_cost /*dev*/, int32_t batch, int32_t channels,
                                  int32_t d_max, int32_t height, int32_t width, int32_t half_precision,
                                  void* grad_x /*dev, out*/, void* grad_y /*dev, out*/, void* stream);

/* nn.AvgPool2d((k, k), stride=(k, k)) of the depth network's two context branches (dfdp/dddnet/dddnet.py:376-385: windows
 * of 32 and of 8 pixels on [B, 128, H/4, W/4] maps): x [planes, height, width] -> out [planes, height / k, width / k],
 * fp32 accumulation and one division by k * k as torch's kernel; k must divide height and width.  half_precision != 0:
 * fp16 tensors (the network runs under autocast), else fp32. */
int sdirt_avg_pool_windows(const void* x /*dev*/, int64_t planes, int32_t height, int32_t width, int32_t k,
                           int32_t half_precision, void* out /*dev*/, void* stream);

/* The same window average READING a pixel-major map (what a channels_last tensor is in memory): x [batch, height, width,
 * channels] -> out [batch, channels, height / k, width / k] (planar, like sdirt_avg_pool_windows).  channels: a multiple of 8 (fp16) / 4 (fp32), at most 2048 / 1024;
 * x 16-byte aligned; SDIRT_ERR_UNSUPPORTED otherwise. */
int sdirt_avg_pool_windows_nhwc(const void* x /*dev*/, int32_t batch, int32_t height, int32_t width, int32_t channels,
                                int32_t k, int32_t half_precision, void* out /*dev*/, void* stream);

/* ---- depth network, inference-time fusions (eval mode; training keeps torch's differentiable ops) ---- */

/* The tail of BasicConv.forward (dfdp/dddnet/dddnet.py:539-543) in eval mode: batch norm with its running statistics
 * (y = gamma * (x - mean) * invstd + beta, invstd = 1 / sqrt(running_var + eps), evaluated in fp32 in torch's term order)
 * followed by ReLU when relu != 0 -- ONE pass, in place, over the convolution's output x = [outer][channels][inner]
 * (planar tensors: outer = batch, inner = the spatial size; channels_last / channels_last_3d tensors: outer = batch x
 * spatial size, inner = 1).  mean / invstd / gamma / beta: dev fp32 [channels].  half_precision != 0: x is fp16. */
int sdirt_bn_relu(void* x /*dev, in place*/, int64_t outer, int32_t channels, int64_t inner, const float* mean /*dev*/,
                  const float* invstd /*dev*/, const float* gamma /*dev*/, const float* beta /*dev*/, int32_t relu,
                  int32_t half_precision, void* stream);

/* Disp + DisparityRegression (dfdp/dddnet/dddnet.py:543-568): cost [B, 1, d_in, h_in, w_in] (the hourglass's output) ->
 * trilinear interpolation to [d_out, h_out, w_out] (align_corners = False, as F.interpolate(x, size)) -> softmin over the
 * d_out shifts -> expectation over the shifts arange(-d_out // 2, d_out // 2) -> disp fp32 [B, 1, h_out, w_out]; all in
 * fp32 (what autocast runs these ops in), one thread per output pixel.  d_in <= 32, d_out <= 64 (the reference: 10 -> 20);
 * SDIRT_ERR_UNSUPPORTED beyond.  half_precision != 0: cost is fp16. */
int sdirt_disparity_regression(const void* cost /*dev*/, int32_t batch, int32_t d_in, int32_t h_in, int32_t w_in,
                               int32_t d_out, int32_t h_out, int32_t w_out, int32_t half_precision,
                               float* disp /*dev, out*/, void* stream);

/* nn.Upsample(mode = 'trilinear', align_corners = True) of Conv2x (dfdp/dddnet/dddnet.py:585, 589) between volumes stored
 * [B, D, H, W, C] (channels_last_3d): x [B, d_in, h_in, w_in, C] -> out [B, d_out, h_out, w_out, C], interpolated in fp32
 * from the tensor's values in torch's term order and rounded once to its type.  16-byte accesses when `channels` is a
 * multiple of 8 (fp16) / 4 (fp32) and both pointers are 16-byte aligned. */
int sdirt_upsample_trilinear_ndhwc(const void* x /*dev*/, int32_t batch, int32_t channels, int32_t d_in, int32_t h_in,
                                   int32_t w_in, int32_t d_out, int32_t h_out, int32_t w_out, int32_t half_precision,
                                   void* out /*dev*/, void* stream);

/* PSFNet tone curves (deeplens/psfnet.py:589-620), elementwise over n floats, in place allowed:
 * mode 0 = degamma(img) (code value in [0,1] -> linear luminance, psfnet.py:600-603),
 * mode 1 = clip(gamma(l), 0, 1) (psfnet.py:617-620 followed by the clip of render, :712). */
int sdirt_tone_curve(const float* in /*dev*/, int64_t n, int32_t mode, float* out /*dev*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDIRT_DP_H */
