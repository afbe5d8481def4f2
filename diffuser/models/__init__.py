from madiff_amd.diffusion import GaussianDiffusion  # noqa: F401
from madiff_amd.modules import (  # noqa: F401
    Conv1dBlock,
    Downsample1d,
    ResidualTemporalBlock,
    SelfAttention,
    SharedConvAttentionDeconv,
    SinusoidalPosEmb,
    TemporalSelfAttention,
    TemporalUnet,
    Upsample1d,
)
from madiff_amd.wrappers import (  # noqa: F401
    ConcatenatedTemporalUnet,
    IndependentTemporalUnet,
    SharedIndependentTemporalUnet,
)
from . import diffusion, helpers, ma_temporal, ma_temporal_wrappers, temporal  # noqa: F401
