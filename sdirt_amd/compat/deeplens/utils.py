"""deeplens.utils -> sdirt_amd.utils."""
from sdirt_amd.utils import set_logger, set_seed  # noqa: F401
