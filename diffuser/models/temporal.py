from madiff_amd.modules import ResidualTemporalBlock, TemporalSelfAttention, TemporalUnet  # noqa: F401
