"""utils/progress.py of the reference."""
from madiff_amd.host import Progress, Silent  # noqa: F401
