"""ops.split_cols: column blocks of the fused time-projection matrix whose gradients the consumers may write in place
(shared buffer, no concatenation in backward).  Plain torch on the CPU: host logic only."""
import torch


class _Consumer(torch.autograd.Function):
    """y = 2 * block; backward writes its gradient into the block's slot when asked to, else returns a fresh tensor."""

    @staticmethod
    def forward(ctx, block, use_slot):
        from madiff_amd import ops
        ctx.slot = ops.grad_slot(block) if use_slot else None
        return block * 2

    @staticmethod
    def backward(ctx, g):
        from madiff_amd import ops
        dst = ops.slot_view(ctx.slot, g.shape[0], g.shape[1])
        if dst is None:
            return g * 2, None
        dst.copy_(g * 2)
        return dst, None


def _run(use):
    from madiff_amd import ops
    torch.manual_seed(0)
    big = torch.randn(5, 24, requires_grad=True)
    sizes = (8, 4, 8)                                   # 4 pad columns stay unused
    outs = ops.split_cols(big, sizes)
    assert all(ops.grad_slot(o) is not None for o in outs)
    w = [torch.randn(5, s) for s in sizes]
    loss = sum((_Consumer.apply(o, u) * wi).sum() for o, u, wi in zip(outs, use, w) if u is not None)
    loss.backward()
    ref = torch.zeros(5, 24)
    off = 0
    for s, u, wi in zip(sizes, use, w):
        if u is not None:
            ref[:, off:off + s] = 2 * wi
        off += s
    assert torch.equal(big.grad, ref)


def test_all_consumers_write_in_place():
    _run([True, True, True])


def test_mixed_and_unused_blocks():
    _run([True, False, True])
    _run([False, False, False])          # nobody uses a slot: the concatenation path
    _run([True, None, False])            # a block without a consumer: zero gradient


def test_no_grad_has_no_slots():
    from madiff_amd import ops
    with torch.no_grad():
        outs = ops.split_cols(torch.randn(3, 8), (4, 4))
    assert all(ops.grad_slot(o) is None for o in outs)


def test_second_backward_gets_a_fresh_buffer():
    from madiff_amd import ops
    big = torch.randn(2, 8, requires_grad=True)
    for _ in range(2):
        outs = ops.split_cols(big, (4, 4))
        (_Consumer.apply(outs[0], True).sum() + _Consumer.apply(outs[1], True).sum()).backward()
    assert torch.equal(big.grad, torch.full((2, 8), 4.0))
