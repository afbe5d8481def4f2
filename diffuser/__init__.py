"""Drop-in ``diffuser`` package name for MADiff's class-path plugin mechanism.

The reference resolves YAML strings such as ``model: "models.SharedConvAttentionDeconv"`` with
``importlib.import_module("diffuser." + module)`` (reference diffuser/utils/config.py:7-21) and its
evaluator unpickles Config objects that hold classes by module path.  This thin package re-exports the
MI355X-native implementations from ``madiff_amd`` under those names; it contains no logic of its own.
"""
from . import models, utils  # noqa: F401
