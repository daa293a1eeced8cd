"""deeplens.monte_carlo -> sdirt_amd.monte_carlo."""
from sdirt_amd.monte_carlo import (assign_points_to_pixels_big_r, assign_points_to_pixels_small_r,  # noqa: F401
                                   forward_integral)
