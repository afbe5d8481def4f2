from madiff_amd.diffusion import Encoder, IdentityEncoder, SMAC3mEncoder, SMAC5m6mEncoder  # noqa: F401
from madiff_amd.host import (  # noqa: F401
    Config,
    Progress,
    Silent,
    apply_dict,
    batch_to_device,
    batchify,
    build_config_from_dict,
    discover_latest_checkpoint_path,
    import_class,
    report_parameters,
    set_device,
    set_seed,
    to_device,
    to_np,
    to_torch,
)
from madiff_amd.training import EMA, Trainer, cycle  # noqa: F401,E402
