"""deeplens.basics -> sdirt_amd.basics (constants, Material, Ray)."""
from sdirt_amd.basics import (DEFAULT_WAVE, DEPTH, EPSILON, GEO_SPP, WAVE_RGB, Material, Ray,  # noqa: F401
                              register_material)
