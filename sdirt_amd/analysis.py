"""The lens report of the reference's fitting script (`psfnet.analysis(...)`, 1_fit_psfnet.py:29-32 ->
Lensgroup.analysis, deeplens/optics.py:1663-1684) and what it stands on: the ray samplers for fans and point
grids (optics.py:217-297, 366-457, 542-594), magnification by ray mapping (optics.py:1237-1270, 1310-1321), RMS
spot radii (optics.py:2103-2140) and the 2-D layout figure with traced rays (optics.py:1686-1882).

Every ray here goes through Lensgroup.trace -- the HIP kernels, with the batch-wide Newton trip counts of the
reference -- on whatever batch shape the caller builds ([M], [spp, M, M]); what this module adds is host
arithmetic on the traced rays and matplotlib (imported when a figure is written).  Functions take the lens as
their first argument; Lensgroup binds them under the reference's method names.
"""
import numpy as np
import torch

from .basics import DEFAULT_WAVE, DEPTH, EPSILON, GEO_SPP, WAVE_RGB, Ray
from .plots import _pyplot


# ------------------------------------------------------------------------------------ samplers
def sample_parallel_2D(lens, R=None, wvln=DEFAULT_WAVE, z=None, view=0.0, M=15, forward=True, entrance_pupil=False):
    """optics.py:217-297: a meridional fan of M parallel rays at `view` degrees.  entrance_pupil: through M
    points across 0.99 of the entrance pupil, started at z = -0.1 when the pupil lies inside the lens; otherwise
    from M points across +-R on z = 0 (forward) or on the sensor (backward, directions mirrored)."""
    sx, cz = float(np.sin(view / 57.3)), float(np.cos(view / 57.3))
    if entrance_pupil:
        pz, pr = lens.entrance_pupil()
        x2 = torch.linspace(-pr, pr, M) * 0.99
        o2 = torch.stack((x2, torch.zeros_like(x2), torch.full_like(x2, pz)), dim=-1)
        d = torch.tensor([sx, 0.0, cz]).expand(M, 3)
        o = o2 - d * ((o2[:, 2] + 0.1) / cz).unsqueeze(-1) if pz > 0 else o2
        return Ray(o, d, wvln, device=lens.device)
    x = torch.linspace(-R, R, M)
    if z is None:
        z = 0 if forward else lens.d_sensor
    o = torch.stack((x, torch.zeros_like(x), torch.full_like(x, z)), dim=-1)
    d = torch.tensor([sx, 0.0, cz] if forward else [-sx, 0.0, -cz]).expand(M, 3)
    return Ray(o, d, wvln, device=lens.device)


def sample_point_source_2D(lens, depth=-1000, view=0, M=9, entrance_pupil=False, wvln=DEFAULT_WAVE):
    """optics.py:366-399: M meridional rays from the point (depth tan(view), 0, depth) to points across 0.99 of
    the entrance pupil (or of the first surface's aperture at z = 0), moved up to 0.1 mm before the first vertex."""
    pz, px = lens.entrance_pupil() if entrance_pupil else (0, lens.surfaces[0].r)
    x2 = torch.linspace(-px, px, M) * 0.99
    o2 = torch.stack((x2, torch.zeros_like(x2), torch.full_like(x2, pz)), dim=1)
    o1 = torch.zeros_like(o2)
    o1[:, 2] = depth
    o1[:, 0] = depth * np.tan(view / 57.3)
    ray = Ray(o1, o2 - o1, wvln=wvln, device=lens.device)
    ray.propagate_to(z=float(lens.surfaces[0].d) - 0.1)
    return ray


def sample_pupil(lens, res=(512, 512), spp=16, num_angle=8, pupilr=None, pupilz=None):
    """optics.py:542-594: [spp, H, W, 3] points on the entrance-pupil disc, an independent set per pixel:
    uniform over the disc, or -- spp a multiple of num_angle and below 10000 -- one point in each of num_angle
    sectors x spp / num_angle equal-area rings.  Drawn on the lens's device generator, as in the reference
    (lens.sample_rng_device = 'cpu': on the CPU generator, then moved to the device)."""
    H, W = res
    if pupilr is None or pupilz is None:
        pupilz, pupilr = lens.entrance_pupil()
    dev = torch.device(getattr(lens, "sample_rng_device", None) or lens.device)
    if spp % num_angle != 0 or spp >= 10000:
        theta = torch.rand((spp, H, W), device=dev) * 2 * np.pi
        r = torch.sqrt(torch.rand((spp, H, W), device=dev) * pupilr ** 2)
        x, y = r * torch.cos(theta), r * torch.sin(theta)
    else:
        xs, ys = [], []
        for sector in range(num_angle):
            for ring in range(spp // num_angle):
                theta = torch.rand((1, H, W), device=dev) * 2 * np.pi / num_angle + sector * 2 * np.pi / num_angle
                r2 = torch.rand((1, H, W), device=dev) * pupilr ** 2 / spp * num_angle \
                    + ring * pupilr ** 2 / spp * num_angle
                r = torch.sqrt(r2)
                xs.append(r * torch.cos(theta))
                ys.append(r * torch.sin(theta))
        x, y = torch.cat(xs, dim=0), torch.cat(ys, dim=0)
    return torch.stack((x, y, torch.full_like(x, pupilz)), -1).to(lens.device)


def point_source_origins(lens, R=None, depth=-10.0, M=11, importance_sampling=False):
    """optics.py:422-442: the [M, M, 3] grid of point sources of sample_point_source: x from -Rw to Rw left to
    right, y from R down to -R (Rw = R x sensor aspect), optionally square-root warped towards the edges."""
    if R is None:
        R = lens.surfaces[0].r
    Rw = R * lens.sensor_res[1] / lens.sensor_res[0]
    x, y = torch.meshgrid(torch.linspace(-1, 1, M), torch.linspace(1, -1, M), indexing="xy")
    if importance_sampling:
        x, y = torch.sqrt(x.abs()) * x.sign(), torch.sqrt(y.abs()) * y.sign()
    x, y = x * Rw, y * R
    return torch.stack((x, y, torch.full_like(x, depth)), -1)


def sample_point_source(lens, R=None, depth=-10.0, M=11, spp=16, fov=10.0, forward=True, pupil=True,
                        wvln=DEFAULT_WAVE, importance_sampling=False):
    """optics.py:402-457: rays [spp, M, M] from an M x M grid of point sources at `depth` to per-source pupil
    samples (sample_pupil); directions are normalised before the Ray constructor normalises them again, as in the
    reference."""
    if not pupil:
        raise Exception("Cone sampling specified by fov has been abandoned. Use pupil sampling instead.")
    o = point_source_origins(lens, R, depth, M, importance_sampling).to(lens.device)
    o = o.unsqueeze(0).repeat(spp, 1, 1, 1)
    d = sample_pupil(lens, res=(M, M), spp=spp) - o
    d = d / torch.linalg.vector_norm(d, ord=2, dim=-1, keepdim=True)
    return Ray(o, d, wvln, device=lens.device)


# ------------------------------------------------------------------------------------ measures
def _weighted_mean_position(p, ra, eps):
    """sum_s p ra / (sum_s ra + eps) over the sample axis 0: p [spp, ..., 2], ra [spp, ...]."""
    return (p * ra.unsqueeze(-1)).sum(0) / ra.sum(0).add(eps).unsqueeze(-1)


def magnification_from_rays(lens, ray, M=21):
    """The arithmetic of optics.py:1249-1270 on a sampled [spp, M, M] bundle: trace it, take the validity-
    weighted mean landing point of every source, and average object x / image x over the upper-left quadrant
    (off the axes, where x vanishes)."""
    x1 = torch.flip(ray.o[..., :2], [1, 2])[0, :, :, 0]
    ray, _, _ = lens.trace(ray)
    x2 = _weighted_mean_position(ray.project_to(lens.d_sensor), ray.ra, EPSILON)[..., 0]
    q = (x1 / x2)[:M // 2, :M // 2]
    return 1 / torch.mean(q[~q.isnan()]).item()


@torch.no_grad()
def calc_magnification3(lens, depth):
    """optics.py:1237-1270: lateral magnification at `depth` from 21 x 21 sources over half the field, 512 rays
    each."""
    M = 21
    ray = sample_point_source(lens, M=M, spp=512, depth=depth, R=-depth * np.tan(lens.hfov) * 0.5, pupil=True)
    mag = magnification_from_rays(lens, ray, M)
    if mag == 0:
        return 1 / (-depth * np.tan(lens.hfov) / lens.r_last)
    return mag


def calc_scale_ray(lens, depth):
    """optics.py:1310-1321: object height per sensor height, 1 / magnification (a tensor of depths maps to a
    tensor)."""
    if isinstance(depth, torch.Tensor) and len(depth.shape) == 1:
        return torch.tensor([1 / calc_magnification3(lens, d) for d in depth])
    return 1 / calc_magnification3(lens, depth)


def _landing_centres(lens, ray):
    """trace `ray` ([spp, H, H]) to the sensor -> validity-weighted mean landing point of every source."""
    ray, _, _ = lens.trace(ray)
    return _weighted_mean_position(ray.project_to(lens.d_sensor), ray.ra, 0.0001)


def _spot_rms(lens, ray, centre, H):
    """optics.py:2124-2135 for one wavelength: RMS distance of the landing points from `centre` (None: from
    their own mean) over the whole field, on the axis and at the field corner.  The index quirk of the
    reference is kept: the on-axis figure sums the squares at [H//2+1, H//2+1] over the weight at [H//2, H//2]."""
    ray, _, _ = lens.trace(ray)
    p, w = ray.project_to(lens.d_sensor), ray.ra
    if centre is None:
        centre = _weighted_mean_position(p, w, 0.0001)
    sq = ((p - centre) * w.unsqueeze(-1)) ** 2 * w.unsqueeze(-1)
    c = H // 2
    return (torch.sqrt(sq.sum() / w.sum()), torch.sqrt(sq[:, c + 1, c + 1, :].sum() / w[:, c, c].sum()),
            torch.sqrt(sq[:, 0, 0, :].sum() / w[:, 0, 0].sum()))


def rms_from_rays(lens, rays, ref_ray=None, H=31):
    """The arithmetic of optics.py:2111-2140 on sampled bundles: `rays` one [spp, H, H] bundle per wavelength,
    `ref_ray` the bundle whose mean landing points are the reference centres (None: every colour about its own).
    -> (rms over the field, on axis, at the field corner), each the mean over the wavelengths."""
    centre = _landing_centres(lens, ref_ray) if ref_ray is not None else None
    per_colour = [_spot_rms(lens, ray, centre, H) for ray in rays]
    return tuple(sum(c[i] for c in per_colour) / len(per_colour) for i in range(3))


@torch.no_grad()
def analysis_rms(lens, depth=DEPTH, ref=True):
    """optics.py:2103-2140: RMS spot radius [mm] over a 31 x 31 field at `depth`, GEO_SPP rays per source and
    wavelength, about the green centres (ref) or each colour's own.  Sampled and traced one bundle at a time in
    the reference's order (green first when ref)."""
    H = 31
    R = lens.sensor_size[0] / 2 * calc_scale_ray(lens, depth)
    sample = lambda w: sample_point_source(lens, M=H, spp=GEO_SPP, depth=depth, R=R, pupil=True, wvln=w)  # noqa: E731
    centre = _landing_centres(lens, sample(DEFAULT_WAVE)) if ref else None
    per_colour = [_spot_rms(lens, sample(w), centre, H) for w in WAVE_RGB]
    return tuple(sum(c[i] for c in per_colour) / len(per_colour) for i in range(3))


def calc_eqfl(lens):
    """optics.py:1119-1124: 35-mm-equivalent focal length."""
    return 21.63 / np.tan(lens.hfov)


# ------------------------------------------------------------------------------------ layout figure
def _stop_marks(ax, s, colour):
    """the four short strokes that mark an aperture stop of semi-diameter r at z = d"""
    d, r = float(s.d), s.r
    for sign in (-1, 1):
        ax.plot([d - 0.05 * r, d, d + 0.05 * r], [sign * r] * 3, colour)
        ax.plot([d] * 3, np.linspace(sign * r, sign * (r + 0.15 * r), 3), colour)


def plot_setup2D(lens, ax=None, fig=None, color="k", with_sensor=True, zmx_format=False, fix_bound=False):
    """optics.py:1792-1882: the lens in the meridional plane -- each refracting surface's profile over its
    aperture (257 samples), stops as wedges, the rims of neighbouring surfaces of one glass element joined, the
    sensor as a line of half-height r_last.  -> (ax, fig)."""
    plt = _pyplot()
    if ax is None and fig is None:
        fig, ax = plt.subplots(figsize=(5, 5))
    air = lambda m: m.A < 1.0003                                   # noqa: E731
    surfs, mats = lens.surfaces, lens.materials
    if len(surfs) == 1:
        _stop_marks(ax, surfs[0], "orange")
    else:
        for i, s in enumerate(surfs):
            if air(mats[i]) and air(mats[i + 1]):
                _stop_marks(ax, s, "orange")
            else:
                r = np.linspace(-s.r, s.r, 257, dtype=np.float32)
                ax.plot(s.surface_with_offset(r, np.zeros_like(r)), r, color)
        front = None
        for i, s in enumerate(surfs):
            if air(mats[i]):
                front = s
                continue
            z0, z1 = float(front.surface_with_offset(front.r, 0.0)), float(s.surface_with_offset(s.r, 0.0))
            zs, xs = ([z0, z0, z1], [front.r, s.r, s.r]) if zmx_format else ([z0, z1], [front.r, s.r])
            ax.plot(zs, xs, color)
            ax.plot(zs, [-v for v in xs], color)
            front = s
        if with_sensor:
            ax.plot([lens.d_sensor, lens.d_sensor], [-lens.r_last, lens.r_last], color)
    plt.xlabel("z [mm]")
    plt.ylabel("r [mm]")
    ax.set_aspect("equal", adjustable="datalim", anchor="C")
    ax.minorticks_on()
    ax.set_xlim(-0.5, 7.5)
    ax.set_ylim(-4, 4)
    ax.autoscale()
    return ax, fig


def plot_raytraces(lens, oss, ax=None, fig=None, color="b-", show=True, p=None, valid_p=None, plot_invalid=True,
                   ra=None):
    """optics.py:1757-1790: the recorded paths `oss` (trace(record=True)) as polylines z -> x; with
    plot_invalid=False only the rays whose weight `ra` is still positive."""
    plt = _pyplot()
    if ax is None and fig is None:
        ax, fig = plot_setup2D(lens)
    else:
        show = False
    for i, path in enumerate(oss):
        pts = np.asarray(path, np.float32).reshape(-1, 3)
        x, z = pts[:, 0], pts[:, 2]
        if p is not None and valid_p is not None and valid_p[i]:
            x, z = np.append(x, float(p[i, 0])), np.append(z, float(p[i, 2]))
        if plot_invalid or float(ra[i]) > 0:
            ax.plot(z, x, color, linewidth=0.8)
    if show:
        plt.show()
    else:
        plt.close()
    return ax, fig


def layout_title(lens):
    """optics.py:1691-1695."""
    head = f"FoV{round(2 * lens.hfov * 57.3, 1)}({int(calc_eqfl(lens))}mm EFL)"
    stop = f"_F/{round(lens.fnum, 2)}" if lens.aper_idx is not None else ""
    return f"{head}{stop}_DIAG{round(lens.r_last * 2, 2)}mm_FocLen{round(lens.foclen, 2)}mm"


def layout_fans(lens, depth=None, M=9, entrance_pupil=True, multi_plot=False):
    """The ray fans of the layout figure (optics.py:1699-1738) and their recorded paths:
    -> list of (wavelength slot, view [deg], ray, oss).  One figure: three views (0, 0.707, 0.99 of the half
    field) in blue / green / red light; multi_plot: seven views for each of the three wavelengths."""
    R = lens.surfaces[0].r
    half = np.rad2deg(lens.hfov)
    if multi_plot:
        jobs = [(i, v, w) for i, w in enumerate(WAVE_RGB) for v in np.linspace(0, half * 0.99, num=7)]
    else:
        jobs = [(i, v, WAVE_RGB[2 - i]) for i, v in enumerate([0, half * 0.707, half * 0.99])]
    out = []
    for slot, view, w in jobs:
        if depth is None:
            ray = sample_parallel_2D(lens, R, w, view=view, M=M, entrance_pupil=entrance_pupil)
        else:
            ray = sample_point_source_2D(lens, depth=depth, view=view, M=M, entrance_pupil=entrance_pupil, wvln=w)
        _, oss = lens.trace2sensor(ray=ray, record=True)
        out.append((slot, view, ray, oss))
    return out


@torch.no_grad()
def plot_setup2D_with_trace(lens, filename, views=[0], M=9, depth=None, entrance_pupil=True, zmx_format=False,
                            plot_invalid=True, multi_plot=False, lens_title=None):
    """optics.py:1686-1742: the layout with traced fans -> <filename>.png (and .svg with multi_plot); returns
    the fans (layout_fans)."""
    plt = _pyplot()
    if lens_title is None:
        lens_title = layout_title(lens)
    fans = layout_fans(lens, depth, M, entrance_pupil, multi_plot)
    if multi_plot:
        fig, axs = plt.subplots(1, 3, figsize=(24, 6))
        fig.suptitle(lens_title)
        for i in range(3):
            plot_setup2D(lens, ax=axs[i], fig=fig, zmx_format=zmx_format)
        for slot, _, ray, oss in fans:
            plot_raytraces(lens, oss, ax=axs[slot], fig=fig, color="rgb"[slot], plot_invalid=plot_invalid,
                           ra=ray.ra.cpu().numpy())
            axs[slot].axis("off")
        fig.savefig(f"{filename}.svg", bbox_inches="tight", format="svg", dpi=600)
        fig.savefig(f"{filename}.png", bbox_inches="tight", format="png", dpi=300)
    else:
        ax, fig = plot_setup2D(lens, zmx_format=zmx_format)
        for slot, _, ray, oss in fans:
            plot_raytraces(lens, oss, ax=ax, fig=fig, color="bgr"[slot], plot_invalid=plot_invalid,
                           ra=ray.ra.cpu().numpy())
        ax.axis("off")
        ax.set_title(lens_title)
        fig.savefig(f"{filename}.png", bbox_inches="tight", format="png", dpi=600)
    plt.close(fig)
    return fans


def analysis(lens, save_name="./test", ks=None, render=False, multi_plot=False, plot_invalid=True, zmx_format=False,
             depth=DEPTH, render_unwarp=False, lens_title=None):
    """optics.py:1663-1684: layout figure with traced fans, PSF map, RMS spot radii (printed as in the
    reference and returned).  render=True (a resolution chart through the lens with PSNR / SSIM, which needs the
    reference's dataset file, OpenCV and scikit-image) is outside this build."""
    if render:
        raise NotImplementedError("analysis(render=True): image rendering with PSNR / SSIM is outside this build")
    plot_setup2D_with_trace(lens, filename=save_name, multi_plot=multi_plot, entrance_pupil=True,
                            plot_invalid=plot_invalid, zmx_format=zmx_format, lens_title=lens_title, depth=depth)
    lens.draw_psf_map(save_name=save_name, ks=ks, depth=depth)
    avg, on_axis, off_axis = analysis_rms(lens, depth=depth)
    print(f"On-axis RMS radius: {round(on_axis.item() * 1000, 3)}um, Off-axis RMS radius: "
          f"{round(off_axis.item() * 1000, 3)}um, Avg RMS spot size (radius): {round(avg.item() * 1000, 3)}um.")
    return avg, on_axis, off_axis
