from madiff_amd.diffusion import GaussianDiffusion  # noqa: F401
from madiff_amd.modules import (  # noqa: F401
    Conv1dBlock,
    Downsample1d,
    ResidualTemporalBlock,
    SharedConvAttentionDeconv,
    SinusoidalPosEmb,
    TemporalSelfAttention,
    TemporalUnet,
    Upsample1d,
)
from . import diffusion, helpers, ma_temporal, temporal  # noqa: F401
