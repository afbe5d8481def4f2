"""utils/launcher_util.py of the reference, the two names the training entry point uses (run_experiment.py /
train.py); the sweep generator is the control plane and stays out of scope."""
from madiff_amd.host import build_config_from_dict, discover_latest_checkpoint_path  # noqa: F401
