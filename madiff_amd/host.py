"""Boundary plumbing the hot path touches in the reference's ``diffuser.utils``:
class-path import + lazy Config (utils/config.py:7-69), array/device moves (utils/arrays.py), progress
reporting (utils/progress.py), seeding (utils/setup.py:13-17), and the in-place inpainting helper
(models/helpers.py:194-206).  ml_logger is optional here: configs are only pickled when it is importable."""
from __future__ import annotations

import collections.abc
import importlib
import random
import time

import numpy as np
import torch

from . import ops

DTYPE = torch.float
DEVICE = "cuda"


def import_class(_class, repo_name="diffuser"):
    if type(_class) is not str:
        return _class
    module_name, class_name = _class.rsplit(".", 1)
    return getattr(importlib.import_module(f"{repo_name}.{module_name}"), class_name)


class Config(collections.abc.Mapping):
    """Lazy constructor: ``Config("models.X", savepath=..., device=..., **kwargs)(*args)`` builds
    ``X(*args, **kwargs)`` and moves it to ``device``."""

    def __init__(self, _class, print_info=False, savepath=None, device=None, **kwargs):
        self._class = import_class(_class)
        self._device = device
        self._dict = dict(kwargs)
        if print_info:
            print(self)
        if savepath is not None:
            try:
                from ml_logger import logger
                logger.save_pkl(self, savepath)
            except ImportError:
                pass

    def __repr__(self):
        return f"\n[utils/config ] Config: {self._class}\n" + "".join(
            f"    {k}: {self._dict[k]}\n" for k in sorted(self._dict))

    def __iter__(self):
        return iter(self._dict)

    def __getitem__(self, item):
        return self._dict[item]

    def __len__(self):
        return len(self._dict)

    def __getattr__(self, attr):
        if attr == "_dict" and "_dict" not in vars(self):
            self._dict = {}
            return self._dict
        try:
            return self._dict[attr]
        except KeyError:
            raise AttributeError(attr)

    def __call__(self, *args, **kwargs):
        instance = self._class(*args, **kwargs, **self._dict)
        if self._device:
            instance = instance.to(self._device)
        return instance


def to_np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else x


def to_torch(x, dtype=None, device=None):
    dtype, device = dtype or DTYPE, device or DEVICE
    if type(x) is dict:
        return {k: to_torch(v, dtype, device) for k, v in x.items()}
    if torch.is_tensor(x):
        return x.to(device).type(dtype)
    return torch.tensor(x, dtype=dtype, device=device)


def to_device(x, device=DEVICE):
    if torch.is_tensor(x):
        return x.to(device)
    if type(x) is dict:
        return {k: to_device(v, device) for k, v in x.items()}
    raise TypeError(f"Unrecognized type in `to_device`: {type(x)}")


def apply_dict(fn, d, *args, **kwargs):
    return {k: fn(v, *args, **kwargs) for k, v in d.items()}


def batchify(batch, device):
    fn = lambda x: to_torch(x[None], device=device)  # noqa: E731
    return {k: (apply_dict(fn, v) if type(v) is dict else fn(v)) for k, v in batch.items()}


def batch_to_device(batch, device="cuda:0"):
    return {k: to_device(v, device) for k, v in batch.items()}


def set_device(device):
    """utils/arrays.py:102-106 without the global default-tensor-type switch (deprecated in this torch)."""
    global DEVICE
    DEVICE = device


def _to_str(num):
    return f"{num / 1e6:.2f} M" if num >= 1e6 else f"{num / 1e3:.2f} k"


def report_parameters(model, topk=10):
    """Parameter count with the top-k tensors listed (utils/arrays.py:131-149); returns the total."""
    counts = {k: p.numel() for k, p in model.named_parameters()}
    total = sum(counts.values())
    print(f"[ utils/arrays ] Total parameters: {_to_str(total)}")
    modules = dict(model.named_modules())
    keys = sorted(counts, key=lambda k: -counts[k])
    for key in keys[:topk]:
        print(" " * 8, f"{key:10}: {_to_str(counts[key])} | {modules[key.rpartition('.')[0]]}")
    rest = keys[topk:]
    print(" " * 8, f"... and {len(rest)} others accounting for {_to_str(sum(counts[k] for k in rest))} parameters")
    return total


def discover_latest_checkpoint_path(checkpoint_dir):
    """Newest ``state_<step>.pt`` in a directory, or None (utils/launcher_util.py:17-29)."""
    import os
    import re
    try:
        names = os.listdir(checkpoint_dir)
    except FileNotFoundError:
        return None
    steps = [int(m.group(1)) for m in (re.fullmatch(r"state_(\d+)\.pt", n) for n in names) if m]
    return os.path.join(checkpoint_dir, f"state_{max(steps)}.pt") if steps else None


def build_config_from_dict(specs, Config=None):
    """A class whose attributes are the experiment spec (utils/launcher_util.py:32-40).  The reference derives the
    class from params_proto's ParamsProto; without that package a plain class carries the same attributes."""
    if Config is None:
        try:
            from params_proto.neo_proto import ParamsProto as _Base
        except ImportError:
            _Base = object

        class Config(_Base):
            pass

    for k, v in specs.items():
        setattr(Config, k, v)
    return Config


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


class Silent:
    def __init__(self, *args, **kwargs):
        pass

    def update(self, *args, **kwargs):
        pass

    def close(self):
        pass


class Progress(Silent):
    """Cheap textual progress: one line every `every` updates (never from inside a captured graph)."""

    def __init__(self, total, name="Progress", every=50, **kwargs):
        self.total, self.name, self.every, self.n, self.t0 = total, name, every, 0, time.time()

    def update(self, description=None, n=1):
        self.n += n
        if self.n % self.every == 0 or self.n == self.total:
            print(f"[ {self.name} ] {self.n} / {self.total} | {time.time() - self.t0:.1f}s", flush=True)


def apply_conditioning(x, conditions):
    """In place, like the reference: x[mask] = cond_x[mask] (fused HIP kernel on the GPU)."""
    m = conditions["masks"].to(device=x.device, dtype=torch.uint8).contiguous()
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
        ops.inpaint_(x, conditions["x"].to(x.device, torch.float32).contiguous(), m)
    else:
        raise RuntimeError("apply_conditioning needs a contiguous fp32 tensor on the GPU (no CPU fallback)")
    return x
