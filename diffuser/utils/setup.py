"""utils/setup.py of the reference: seeding (the argument parser is the control plane, out of scope)."""
from madiff_amd.host import set_seed  # noqa: F401
