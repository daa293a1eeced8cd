"""deeplens.psfnet_arch -> sdirt_amd.psfnet_arch."""
from sdirt_amd.psfnet_arch import MLP, MLPConv, initialize_weights  # noqa: F401
