"""utils/training.py of the reference."""
from madiff_amd.training import EMA, Trainer, cycle  # noqa: F401
