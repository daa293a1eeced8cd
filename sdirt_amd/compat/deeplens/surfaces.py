"""deeplens.surfaces -> sdirt_amd.surfaces."""
from sdirt_amd.surfaces import Aspheric  # noqa: F401
