"""utils/data_encoder.py of the reference."""
from madiff_amd.diffusion import Encoder, IdentityEncoder, SMAC3mEncoder, SMAC5m6mEncoder  # noqa: F401
