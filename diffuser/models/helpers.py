from madiff_amd.diffusion import WeightedStateL2  # noqa: F401
from madiff_amd.modules import Conv1dBlock, Downsample1d, SelfAttention, SinusoidalPosEmb, Upsample1d  # noqa: F401
from madiff_amd.host import apply_conditioning  # noqa: F401

Losses = {"state_l2": WeightedStateL2}
