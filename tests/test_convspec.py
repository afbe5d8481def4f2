"""ConvSpec is compared by value: a deep-copied model (the EMA model of utils/training.py:57) must keep the fused paths."""
import copy
import pickle

import pytest
import torch


def test_convspec_value_semantics():
    from madiff_amd import ops
    s = ops.ConvSpec("conv", 5, 1, 2, 0)
    assert s == ops.CONV5 and not (s != ops.CONV5) and hash(s) == hash(ops.CONV5)
    assert s != ops.CONV1 and ops.DOWN != ops.UP
    assert copy.deepcopy(ops.CONV5) is ops.CONV5
    assert pickle.loads(pickle.dumps(ops.CONV5)) == ops.CONV5


def test_deepcopied_model_keeps_conv5_blocks():
    import diffuser.models as DM
    net = DM.SharedConvAttentionDeconv(horizon=8, transition_dim=6, dim=32, dim_mults=(1, 2), n_agents=2,
                                       returns_condition=True, condition_dropout=0.25)
    twin = copy.deepcopy(net)
    from madiff_amd import ops
    for b in twin.net.rtb_blocks():
        assert b.blocks[0].spec == ops.CONV5 and b.blocks[1].spec == ops.CONV5


@pytest.mark.gpu
def test_deepcopied_model_takes_the_fused_conv_gn_launch():
    """Sampling with copy.deepcopy(model) (what Trainer's EMA model is) launches as many fused conv+GroupNorm kernels as
    the original does."""
    import diffuser.models as DM
    from madiff_amd import ops
    ops.set_precision("bf16")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DM.SharedConvAttentionDeconv(horizon=24, transition_dim=16, dim=128, dim_mults=(1, 4, 8), n_agents=3,
                                       returns_condition=True, condition_dropout=0.25).to(dev)
    twin = copy.deepcopy(net)
    real = ops._conv_block_fused
    count = {"fused": 0, "not": 0}

    def spy(*a, **k):
        r = real(*a, **k)
        count["fused" if r is not None else "not"] += 1
        return r

    x = torch.randn(8, 24, 3, 16, device=dev)
    t = torch.randint(0, 50, (8,), device=dev)
    ret = torch.rand(8, 1, 3, device=dev)
    ops._conv_block_fused = spy
    try:
        with torch.no_grad():
            y0 = net(x, t, returns=ret, use_dropout=False)
            first = dict(count)
            count["fused"] = count["not"] = 0
            y1 = twin(x, t, returns=ret, use_dropout=False)
    finally:
        ops._conv_block_fused = real
    assert first["fused"] > 0 and count == first, (first, count)
    assert torch.equal(y0, y1)
