"""The diffuser.utils names the training / evaluation entry points import: same names, same
argument meaning (utils/config.py, utils/arrays.py:131-149, utils/launcher_util.py:17-40)."""
import importlib

import torch


def test_submodules_resolve():
    for mod, names in {"config": ["Config", "import_class"], "arrays": ["to_np", "to_torch", "report_parameters", "batchify"],
                       "launcher_util": ["build_config_from_dict", "discover_latest_checkpoint_path"],
                       "progress": ["Progress", "Silent"], "setup": ["set_seed"], "training": ["Trainer", "EMA", "cycle"],
                       "data_encoder": ["IdentityEncoder", "SMAC3mEncoder", "SMAC5m6mEncoder"]}.items():
        m = importlib.import_module("diffuser.utils." + mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    import diffuser.utils as U
    assert U.import_class("models.GaussianDiffusion").__name__ == "GaussianDiffusion"
    assert U.import_class(dict) is dict


def test_report_parameters(capsys):
    import diffuser.utils as U
    net = torch.nn.Sequential(torch.nn.Linear(1000, 2000), torch.nn.Linear(3, 4))
    n = U.report_parameters(net, topk=1)
    out = capsys.readouterr().out
    assert n == 1000 * 2000 + 2000 + 12 + 4
    assert "Total parameters: 2.00 M" in out and "0.weight" in out and "3 others accounting for 2.02 k" in out


def test_checkpoint_discovery_and_spec_class(tmp_path):
    import diffuser.utils as U
    assert U.discover_latest_checkpoint_path(str(tmp_path / "absent")) is None
    assert U.discover_latest_checkpoint_path(str(tmp_path)) is None
    for n in ("state_100.pt", "state_2000.pt", "state_30.pt", "other.pt", "state_x.txt"):
        (tmp_path / n).write_bytes(b"")
    assert U.discover_latest_checkpoint_path(str(tmp_path)).endswith("state_2000.pt")
    cfg = U.build_config_from_dict({"horizon": 4, "loader": "datasets.SequenceDataset"})
    assert cfg.horizon == 4 and cfg.loader == "datasets.SequenceDataset"

    class Mine:
        seed = 1
    assert U.build_config_from_dict({"horizon": 8}, Mine) is Mine and Mine.horizon == 8 and Mine.seed == 1


def test_config_builds_lazily():
    import diffuser.utils as U
    c = U.Config(torch.nn.Linear, in_features=3, out_features=5)
    assert dict(c) == {"in_features": 3, "out_features": 5} and c.in_features == 3
    assert c().weight.shape == (5, 3)


def test_tune_cache_round_trip(tmp_path):
    """Kernel-variant choices survive a save / load (keys are tuples of layer geometry, values ints / pairs / markers)."""
    from madiff_amd import ops
    saved = {p: dict(d) for p, d in ops._TUNE_BY_PRECISION.items()}
    try:
        ops._TUNE_BY_PRECISION.setdefault("bf16", {}).update({("fwd", 6144, 1024, 5, True): 15, ("wgrad", 6144, 64): (8, 1),
                                                              ("wt", 12, 3): "uncovered"})
        path = str(tmp_path / "tune.json")
        ops.save_tune_cache(path)
        for d in ops._TUNE_BY_PRECISION.values():
            d.clear()
        assert ops.load_tune_cache(path) >= 3
        d = ops._TUNE_BY_PRECISION["bf16"]
        assert d[("fwd", 6144, 1024, 5, True)] == 15 and d[("wgrad", 6144, 64)] == (8, 1) and d[("wt", 12, 3)] == "uncovered"
    finally:
        for p in list(ops._TUNE_BY_PRECISION):
            ops._TUNE_BY_PRECISION[p].clear()
            ops._TUNE_BY_PRECISION[p].update(saved.get(p, {}))
