from madiff_amd.modules import SharedConvAttentionDeconv  # noqa: F401
