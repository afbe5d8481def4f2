"""utils/config.py of the reference: class-path import and the lazy constructor."""
from madiff_amd.host import Config, import_class  # noqa: F401
