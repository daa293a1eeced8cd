"""deeplens.render_psf -> sdirt_amd.render_psf."""
from sdirt_amd.render_psf import (local_dp_psf_render, local_psf_render, local_psf_render_fast,  # noqa: F401
                                  local_psf_render_high_res, render_psf, render_psf_map)
