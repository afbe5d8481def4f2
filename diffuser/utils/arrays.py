"""utils/arrays.py of the reference: array/device moves and parameter counting."""
from madiff_amd.host import (  # noqa: F401
    DEVICE,
    DTYPE,
    apply_dict,
    batch_to_device,
    batchify,
    report_parameters,
    set_device,
    to_device,
    to_np,
    to_torch,
)
