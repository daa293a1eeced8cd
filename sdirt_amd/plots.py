"""The reference's plotting callers of the PSF path (deeplens/optics.py:1884-1956, 2041-2067):
draw_psf_map, draw_psf_radial, draw_mtf -- same arguments, same files written, and the plotted data
returned as well (the reference returns None).  The PSFs come from the HIP path (Lensgroup.psf_map /
psf_rgb / psf_diff); what is here is the arithmetic between them and the figure, and matplotlib
(imported when a figure is actually written).  torchvision's make_grid / save_image, which
draw_psf_radial uses in the reference, are restated in numpy (tile_grid, save_normalised).
"""
import numpy as np
import torch

from .basics import DEFAULT_WAVE, DEPTH, EPSILON, GEO_SPP


def _pyplot():
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    return plt


def tile_grid(tiles, nrow, padding=1, pad_value=0.0):
    """torchvision.utils.make_grid for equally sized [C,h,w] tiles: `nrow` tiles per row, each behind
    `padding` pixels of `pad_value` on its top and left, one more band at the bottom and right."""
    tiles = [np.asarray(t.detach().cpu() if torch.is_tensor(t) else t, np.float32) for t in tiles]
    c, h, w = tiles[0].shape
    cols = min(nrow, len(tiles))
    rows = -(-len(tiles) // cols)
    out = np.full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), pad_value, np.float32)
    for i, t in enumerate(tiles):
        y, x = (i // cols) * (h + padding) + padding, (i % cols) * (w + padding) + padding
        out[:, y:y + h, x:x + w] = t
    return out


def save_normalised(img, path):
    """torchvision.utils.save_image(img, path, normalize=True): shifted and scaled to [0, 1] over the whole
    image (max - min + 1e-5 in the denominator), rounded to 8 bits."""
    img = np.asarray(img, np.float32)
    img = (img - img.min()) / (img.max() - img.min() + 1e-5)
    rgb = np.clip(img * 255 + 0.5, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    _pyplot().imsave(path, rgb)
    return rgb


@torch.no_grad()
def psf_map_normalised(lens, grid=9, depth=DEPTH, ks=51, log_scale=False):
    """optics.py:1888-1905: the RGB PSF map at GEO_SPP * 30 samples, every ks x ks field divided by its own
    maximum (over the three colours), optionally log(map + 1e-3).  -> [3, grid*ks, grid*ks] tensor."""
    m = lens.psf_map(depth=depth, grid=grid, ks=ks, spp=GEO_SPP * 30, center=True)
    t = m.view(3, grid, ks, grid, ks)
    t /= t.amax(dim=(0, 2, 4), keepdim=True)
    if log_scale:
        m = torch.log(m + 1e-3)
    return m


@torch.no_grad()
def draw_psf_map(lens, grid=9, depth=DEPTH, ks=51, log_scale=False, quater=False, save_name=None):
    """optics.py:1884-1931 -> the [grid*ks, grid*ks, 3] array shown (`quater` is accepted and, as in the
    reference, only matters to a branch that is switched off there)."""
    img = psf_map_normalised(lens, grid, depth, ks, log_scale).permute(1, 2, 0).cpu().numpy()
    plt = _pyplot()
    plt.figure(figsize=(3000, 3000), dpi=1)
    plt.imshow(img)
    plt.axis("off")
    plt.tight_layout(pad=0)
    plt.savefig(f"./psf{-depth}mm_left.png" if save_name is None else f"{save_name}_psf{-depth}mm_left.png")
    plt.close()
    return img


@torch.no_grad()
def psf_radial(lens, M=3, depth=DEPTH, ks=51, log_scale=False):
    """optics.py:1938-1953: M fields on the 45-degree diagonal (x = y = 0 ... 1), psf_rgb at 4096 samples,
    each divided by its maximum; log_scale: log(psf + 1e-9) stretched to [0, 1].  -> list of [3,ks,ks]."""
    x = torch.linspace(0, 1, M)
    points = torch.stack((x, x, torch.full_like(x, depth)), dim=-1)
    out = []
    for i in range(M):
        psf = lens.psf_rgb(points=points[i], ks=ks, center=True, spp=4096)
        psf = psf / psf.max()
        if log_scale:
            psf = torch.log(psf + EPSILON)
            psf = (psf - psf.min()) / (psf.max() - psf.min())
        out.append(psf)
    return out


@torch.no_grad()
def draw_psf_radial(lens, M=3, depth=DEPTH, ks=51, log_scale=False, save_name="./psf_radial.png"):
    """optics.py:1934-1956 -> the list of PSFs drawn."""
    psfs = psf_radial(lens, M, depth, ks, log_scale)
    save_normalised(tile_grid(psfs, nrow=M, padding=1, pad_value=0.0), save_name)
    return psfs


@torch.no_grad()
def mtf_curves(lens, relative_fov=(0.0, 0.7, 1.0), wvlns=DEFAULT_WAVE, depth=DEPTH):
    """optics.py:2048-2056: per wavelength and field (fov, fov, depth): psf_diff at ks 256 and its MTF.
    -> list of dicts(wvln, fov, fov_deg, psf, freq, tangential, sagittal)."""
    fovs = [relative_fov] if isinstance(relative_fov, float) else list(relative_fov)
    waves = [wvlns] if isinstance(wvlns, float) else list(wvlns)
    out = []
    for wvln in waves:
        for fov in fovs:
            psf = lens.psf_diff(points=torch.Tensor([fov, fov, depth]), wvln=wvln, ks=256)
            freq, tan, sag = lens.psf2mtf(psf)
            out.append(dict(wvln=wvln, fov=fov, fov_deg=round(fov * lens.hfov * 57.3, 1), psf=psf,
                            freq=freq, tangential=tan, sagittal=sag))
    return out


@torch.no_grad()
def draw_mtf(lens, relative_fov=[0.0, 0.7, 1.0], save_name="./mtf.png", wvlns=DEFAULT_WAVE, depth=DEPTH):
    """optics.py:2041-2067 -> the curves drawn (mtf_curves)."""
    if save_name[-4:] != ".png":
        save_name += ".png"
    curves = mtf_curves(lens, relative_fov, wvlns, depth)
    n_fov = len(curves) // max(1, len([wvlns] if isinstance(wvlns, float) else list(wvlns)))
    plt = _pyplot()
    plt.figure(figsize=(6, 6))
    for i, c in enumerate(curves):
        colour = "rgb"[(i % n_fov) % 3]
        plt.plot(c["freq"], c["tangential"], colour, label=f"{c['fov_deg']}(deg)-Tangential")
        plt.plot(c["freq"], c["sagittal"], colour, label=f"{c['fov_deg']}(deg)-Sagittal", linestyle="--")
    plt.legend()
    plt.xlabel("Spatial Frequency [cycles/mm]")
    plt.ylabel("MTF")
    plt.savefig(f"{save_name}", bbox_inches="tight", format="png", dpi=300)
    plt.close()
    return curves
