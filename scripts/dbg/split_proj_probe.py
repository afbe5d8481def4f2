"""One guided forward: split projection (R + v_t) against the one-GEMM form, same inputs (bf16 build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from madiff_amd import modules as M, ops
import test_gpu_parity as P
from oracle import madiff_oracle as O
ops.set_precision("bf16")
dev = torch.device("cuda:0")
A, Tt, Fd, hh = P.SHAPES["spread"]
for width in (32, 128):
    dif, act = P.build("spread", width, 12)
    net = dif.model
    B = 5
    x = torch.randn(B, Tt, A, Fd, device=dev)
    ret = torch.rand(B, 1, A, device=dev)
    t = torch.full((B,), 7, dtype=torch.int64, device=dev)
    with torch.no_grad():
        xcl = ops.pack_traj(x, ops.pad32(Fd))
        M.PROJ_SPLIT_MIN_SEQ = 0
        R = net.proj_static(ret, B)
        ec0, eu0 = net.forward_guided_pair(xcl, B, Tt, t, ret)
        ec1, eu1 = net.forward_guided_pair(xcl, B, Tt, t, ret, proj_R=R)
        # the projection itself
        both = net.net.embed_fused(t, ret, A, None, False, act=True, twin=True)
        blocks, layers, n_attn = net._time_layers()
        full = ops.conv_gemm(both, [l.weight for l in layers], [l.bias for l in layers], M.LINEAR, 1, perm="f32")
        v = net.proj_step(t)
        off = sum(b.out_channels for b in blocks)
        tot = off + sum(2 * net.self_attn[i].cq + net.self_attn[i].cv for i in range(n_attn))
        d = (R + v)[:, :off] - full[:, :off]
        print(width, "split used", net.last_proj_split, "| block columns: max|R+v-full|", float(d.abs().max()), "max|full|", float(full[:, :off].abs().max()))
        re = lambda a, b: float((a - b).abs().max() / b.abs().max())
        print("   eps cond rel", re(ec1, ec0), "uncond rel", re(eu1, eu0))
