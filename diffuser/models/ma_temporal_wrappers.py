from madiff_amd.wrappers import (  # noqa: F401
    ConcatenatedTemporalUnet,
    IndependentTemporalUnet,
    SharedIndependentTemporalUnet,
)
