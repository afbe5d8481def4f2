#!/usr/bin/env python
"""Benchmark of the MADiff hot path on MI355X: denoiser training step (primary metric) + K-step DDPM sampling.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): MADiff-D MPE simple_tag, 4 agents, horizon 24, F=16, action dim 5, K=200 DDPM
steps, dim 128, dim_mults [1,4,8], returns-conditioned (w=1.2), bf16 build, synthetic batch 256 PER GPU (weak scaling).
One "step" = one optimizer step on one batch of 256 trajectory windows: GaussianDiffusion.loss (add-noise + inpaint,
denoiser forward, masked MSE, inverse-dynamics loss) + backward + gradient all-reduce (N>1) + fused Adam + EMA.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The headline workload is BASELINE.json configs[1] ("tag"); --workload selects the other named shapes for
# the per-shape table in DESIGN.md.  T = horizon + history_horizon; gf = dense forward GFLOP per sample.
WORKLOADS = {
    "tag": dict(A=4, T=24, HH=0, F_OBS=16, ACT=5, DISCRETE=False, NUM_ACTIONS=0, gf=3.295,
                desc="MADiff-D MPE simple_tag denoiser train step: 4 agents, T=24, F=16, act 5 (BASELINE.json configs[1])"),
    "spread": dict(A=3, T=24, HH=0, F_OBS=18, ACT=5, DISCRETE=False, NUM_ACTIONS=0, gf=2.471,
                   desc="MPE simple_spread: 3 agents, T=24, F=18, act 5 (BASELINE.json configs[0] shape)"),
    "ant": dict(A=4, T=28, HH=18, F_OBS=54, ACT=2, DISCRETE=False, NUM_ACTIONS=0, gf=3.848,
                desc="MA-Mujoco 4ant: 4 agents, horizon 10 + history 18 = T 28, F=54 (declared synthetic), act 2 "
                     "(BASELINE.json configs[2] shape)"),
    "smac": dict(A=8, T=24, HH=20, F_OBS=88, ACT=1, DISCRETE=True, NUM_ACTIONS=14, gf=6.614,
                 desc="SMAC 8m: 8 agents, horizon 4 + history 20 = T 24, F=88, 14 discrete actions with legal-action "
                      "masking (BASELINE.json configs[3] shape)"),
}
A, T, HH, F_OBS, ACT, DISCRETE, NUM_ACTIONS = 4, 24, 0, 16, 5, False, 0
K_DIFF, BATCH = 200, 256
FWD_GF_PER_SAMPLE = 3.295          # SURVEY.md section 8d, A=4, T=24, F=16 (dense fwd FLOPs / sample)
WORKLOAD_DESC = WORKLOADS["tag"]["desc"]
MFMA_BF16_PEAK_TF = 2500.0         # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def select_workload(name, batch):
    global A, T, HH, F_OBS, ACT, DISCRETE, NUM_ACTIONS, FWD_GF_PER_SAMPLE, WORKLOAD_DESC, BATCH
    w = WORKLOADS[name]
    A, T, HH, F_OBS, ACT, DISCRETE, NUM_ACTIONS = (w[k] for k in ("A", "T", "HH", "F_OBS", "ACT", "DISCRETE", "NUM_ACTIONS"))
    FWD_GF_PER_SAMPLE, WORKLOAD_DESC, BATCH = w["gf"], w["desc"], batch


def synthetic_batch(batch, device, seed):
    """SURVEY.md 8d synthetic inputs: x ~ U(-1,1) (discrete: column 0 = action id), history + current observation
    conditioned (all agents visible: t < hh + 1), loss mask 1 for t >= hh and 0 at t = hh, legal actions Bernoulli(0.7)
    with the taken action forced legal."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, T, A, ACT + F_OBS, generator=g) * 2 - 1
    masks = torch.zeros(batch, T, A, F_OBS, dtype=torch.bool)
    masks[:, :HH + 1] = True
    loss_masks = torch.zeros(batch, T, A, 1, dtype=torch.float64)
    loss_masks[:, HH + 1:] = 1.0
    b = {"x": x, "cond": {"x": x[..., ACT:].clone(), "masks": masks}, "loss_masks": loss_masks,
         "attention_masks": torch.ones(batch, T, A, 1, dtype=torch.float64),
         "returns": torch.rand(batch, 1, A, generator=g)}
    if DISCRETE:
        act = torch.randint(0, NUM_ACTIONS, (batch, T, A), generator=g)
        x[..., 0] = act.float()
        legal = (torch.rand(batch, T, A, NUM_ACTIONS, generator=g) < 0.7).float()
        legal.scatter_(-1, act[..., None], 1.0)
        b["legal_actions"] = legal
    mv = lambda v: {k: mv(u) for k, u in v.items()} if isinstance(v, dict) else v.to(device)  # noqa: E731
    return mv(b)


def build(device):
    import diffuser.models as DM
    net = DM.SharedConvAttentionDeconv(horizon=T, transition_dim=F_OBS, dim=128, history_horizon=HH, dim_mults=(1, 4, 8),
                                       n_agents=A, returns_condition=True, condition_dropout=0.25)
    dif = DM.GaussianDiffusion(net, n_agents=A, horizon=T - HH, history_horizon=HH, observation_dim=F_OBS, action_dim=ACT,
                               discrete_action=DISCRETE, num_actions=NUM_ACTIONS,
                               n_timesteps=K_DIFF, returns_condition=True, condition_guidance_w=1.2, hidden_dim=256,
                               clip_denoised=True, predict_epsilon=True, share_inv=True).to(device)
    for a in net.self_attn:
        a.gamma.data.fill_(0.5)        # the reference's init (0) would make attention an identity
    return dif


KERNEL_DESC = {
    "conv_taps_bf16_kernel": "conv_taps_bf16_kernel<WM, WT, D, GNB> (all-taps k5 implicit-GEMM convolution, EVERY launch of the "
                             "template in the step: forward with the fused GroupNorm+Mish epilogue, plain data gradient, and the "
                             "<GNB> data gradients that run the preceding block's GroupNorm+Mish backward in their epilogue; "
                             "by_variant splits the two timing labels)",
    "conv_taps_bf16_kernel<GNB>": "conv_taps_bf16_kernel<.., GNB = true> (k5 data gradient with the preceding block's GroupNorm+Mish "
                                  "backward as its epilogue)",
    "gemm_fwd_bf16_kernel": "gemm_fwd_bf16_kernel<OutT, BM, BN, WT, D> (LDS-DMA implicit-GEMM conv/linear: forward + "
                            "data-gradient launches of the 1x1 / strided / linear layers, all tile instantiations)",
    "gemm_wgrad_taps_bf16_kernel": "gemm_wgrad_taps_bf16_kernel<BO, BC, 5, NQ, D> (all-taps k5 weight gradient)",
    "gemm_wgrad_bf16_kernel": "gemm_wgrad_bf16_kernel<BO, BC> (per-tap weight gradient)",
}


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def load_traffic(kernel, launches_per_step):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
    in separate runs of this same command by scripts/pmc_traffic.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for 16-byte-per-lane reads on gfx950).  None when no PMC summary has been recorded for the kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None
    return {"bytes_per_launch": rec["bytes_per_launch"], "fetch_bytes": rec["fetch_bytes"], "write_bytes": rec["write_bytes"],
            "launches_per_step_in_pmc_run": rec["launches_per_step"], "source": "profiles/pmc_traffic.json (" + rec["command"] + ")"}


def cpu_baseline(seconds_budget=30.0):
    """The oracle (plain PyTorch fp32 restatement of the reference path) on this box's host cores, on a bounded
    sample of the same workload: one optimizer step (fwd + bwd + Adam) on 32 of the 256 windows."""
    from oracle import madiff_oracle as O
    cores = len(os.sched_getaffinity(0))
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    dcfg = O.DenoiserCfg(horizon=T, transition_dim=F_OBS, dim=128, dim_mults=(1, 4, 8), n_agents=A,
                         returns_condition=True, condition_dropout=0.25, history_horizon=HH)
    c = O.DiffusionCfg(A, T - HH, HH, F_OBS, ACT, K_DIFF, hidden_dim=256, returns_condition=True,
                       discrete_action=DISCRETE, num_actions=NUM_ACTIONS)
    shapes = {**{"model." + k: v for k, v in O.denoiser_param_shapes(dcfg).items()}, **O.inv_param_shapes(c)}
    params = {k: O.closed_form_param(k, s).requires_grad_(True) for k, s in shapes.items()}
    live = [p for k, p in params.items() if not k.startswith("model.self_attn.3.")]
    opt = torch.optim.Adam(live, lr=2e-4)
    sched = O.DDPMSchedule(K_DIFF)
    nb = 64
    batch = O.synthetic_batch(nb, T, A, F_OBS, ACT, HH, discrete=DISCRETE, num_actions=NUM_ACTIONS, seed=0)
    g = torch.Generator().manual_seed(0)

    def one_step():
        t = torch.randint(0, K_DIFF, (nb,), generator=g)
        noise = torch.randn(nb, T, A, F_OBS, generator=g)
        dm = (torch.rand(nb, A, 1, generator=g) < 0.75).float()
        opt.zero_grad()
        loss, _, _ = O.diffusion_loss(params, dcfg, c, sched, batch["x"], batch["cond"], batch["loss_masks"], t, noise,
                                      batch["returns"], dm, **({"legal_actions": batch["legal_actions"]} if DISCRETE else {}))
        loss.backward()
        opt.step()

    one_step()                                   # warm-up (thread pool, oneDNN primitive caches)
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < seconds_budget * 0.5 and n < 64):
        one_step()
        n += 1
    dt = time.time() - t0
    return {"value": nb * n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} optimizer steps (fwd+bwd+Adam, fp32, torch CPU oracle) on micro-batches of {nb} of the "
                      f"{BATCH} windows (same shape), after 1 warm-up step: {dt:.1f} s"}


def cpu_baseline_sample(episodes=64, steps=5, seconds_budget=12.0):
    """BASELINE.md section 3 "Sample": the oracle's guided K-step loop (2 forwards per step, fp32, torch CPU) on this box's
    host cores, timed over >= `steps` denoising steps on `episodes` episodes of the current sampling shape and
    EXTRAPOLATED to K steps (a full K=200 run takes tens of minutes on the host)."""
    from oracle import madiff_oracle as O
    cores = len(os.sched_getaffinity(0))
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    dcfg = O.DenoiserCfg(horizon=T, transition_dim=F_OBS, dim=128, dim_mults=(1, 4, 8), n_agents=A,
                         returns_condition=True, condition_dropout=0.25, history_horizon=HH)
    c = O.DiffusionCfg(A, T - HH, HH, F_OBS, ACT, K_DIFF, hidden_dim=256, returns_condition=True,
                       discrete_action=DISCRETE, num_actions=NUM_ACTIONS)
    shapes = {"model." + k: v for k, v in O.denoiser_param_shapes(dcfg).items()}
    params = {k: O.closed_form_param(k, s) for k, s in shapes.items()}
    den, _ = O.split_params(params)
    sched = O.DDPMSchedule(K_DIFF)
    b = O.synthetic_batch(episodes, T, A, F_OBS, ACT, HH, discrete=DISCRETE, num_actions=NUM_ACTIONS, seed=0)
    g = torch.Generator().manual_seed(0)
    x = 0.5 * torch.randn(episodes, T, A, F_OBS, generator=g)
    ts_all = sched.timesteps.tolist()

    def one(i, x):
        with torch.no_grad():
            x = O.apply_conditioning(x, b["cond"])
            ts = torch.full((episodes,), ts_all[i], dtype=torch.long)
            eps = O.guided_eps(den, dcfg, c, x, ts, b["returns"])
            return sched.step(eps, ts_all[i], x, torch.randn(x.shape, generator=g))

    x = one(0, x)                                   # warm-up
    t0 = time.time()
    n = 0
    while n < steps or (time.time() - t0 < seconds_budget and n < K_DIFF - 2):
        x = one(1 + n, x)
        n += 1
    dt = time.time() - t0
    return {"value": episodes / (dt / n * K_DIFF), "unit": "trajectories/s", "cores": cores, "kind": "port",
            "extrapolated": True,
            "sample": f"{n} guided denoising steps (2 forwards + posterior each, fp32, torch CPU oracle) on {episodes} "
                      f"episodes of the same shape after 1 warm-up step: {dt:.1f} s, extrapolated x{K_DIFF}/{n} to K={K_DIFF}"}


def launch_ranks(n, argv, script=None, check_devices=True, port=None):
    """`python bench.py --gpus N` without a launcher around it: start N ranks (one per GPU) as children of THIS process,
    which has made no GPU call yet (torch.cuda.device_count() does not initialise the runtime), through
    `python -m torch.distributed.run` on 127.0.0.1, and return the job's exit code.  Fewer visible devices than ranks is
    an error, not a silent 1-rank run.  (The reference has no counterpart: run_scripts/train.py:223,233 pins one GPU.)"""
    import socket
    import subprocess
    if check_devices:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} asked for {n} ranks but only {have} GPU(s) are visible; refusing to run a "
                  f"smaller job under that label", file=sys.stderr, flush=True)
            return 2
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


SAMPLE_GRAPH_DESC = ("the whole K-step loop (2B-batch forward, guidance mix, posterior step with device-side Philox noise, "
                     "inpainting) captured once and replayed as ONE hipGraph launch per rank after one eager step")


def dp1_ms_per_step(args):
    """ms/step of the DATA-PARALLEL code path at one rank (1-rank RCCL group: bucket events, all-reduce launches on the
    communication stream, eager fused Adam behind them), measured by a child process running this same file with
    MADIFF_FORCE_DP=1 after the headline measurement: what the N=1 -> N=2 step of the scaling table pays before any
    inter-GPU traffic.  None when the child fails."""
    import subprocess
    env = dict(os.environ, MADIFF_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 200))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-sample", "--no-cpu", "--workload", args.workload, "--batch", str(args.batch)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)["ms_per_step"]
    except Exception as e:                           # noqa: BLE001
        print(f"bench.py: one-rank data-parallel rehearsal failed ({e!r})", file=sys.stderr, flush=True)
        return None


def same_box_matmul(device, n_seq=1024, t=6, cin=1024, cout=1024, iters=20):
    """The vendor library on this box beside the kernel: torch.matmul (hipBLASLt) in bf16 on the im2col-EQUIVALENT dense
    problem [n_seq*t, 5*cin] x [5*cin, cout] of the largest k5 layer of the headline workload, and this repository's
    all-taps kernel on the real layer (plain convolution + bias, no GroupNorm epilogue, so both do the same arithmetic;
    the library is not charged for the im2col it would need).  Both timed as `iters` launches inside one graph replay."""
    from madiff_amd import ops
    M, K = n_seq * t, 5 * cin
    flops = 2.0 * M * K * cout

    def graph_us(fn):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (3 * iters)

    with torch.no_grad():
        a = torch.randn(M, K, device=device, dtype=torch.bfloat16)
        b = torch.randn(K, cout, device=device, dtype=torch.bfloat16)
        lib_us = graph_us(lambda: torch.matmul(a, b))
        del a, b
        x = torch.randn(M, cin, device=device).to(torch.bfloat16)
        w = torch.randn(cout, cin, 5, device=device) / (5 * cin) ** 0.5
        bias = torch.zeros(cout, device=device)
        own_us = graph_us(lambda: ops.conv_gemm(x, [w], [bias], ops.CONV5, t))
    return {"problem": f"k5 conv {cin}->{cout}, {n_seq} sequences x T={t}: [{M}, {K}] x [{K}, {cout}] bf16",
            "torch_matmul_us": lib_us, "torch_matmul_tflops": flops / lib_us / 1e6,
            "conv_taps_us": own_us, "conv_taps_tflops": flops / own_us / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-sample", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dp1", action="store_true", help="skip the one-rank data-parallel rehearsal (extra.dp1_ms_per_step)")
    ap.add_argument("--no-graph", action="store_true", help="single-GPU: issue the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--sample-batch", type=int, default=1024,
                    help="episodes sampled per GPU (BASELINE.json configs[4]: 8192 episodes over 8 GPUs = 1024 per GPU)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="tag",
                    help="named shape (default: BASELINE.json configs[1], the headline workload)")
    ap.add_argument("--sample-workload", choices=sorted(WORKLOADS), default=None,
                    help="shape of the sampling leg (default: spread on the headline run = BASELINE.json configs[4], "
                         "else the training workload)")
    ap.add_argument("--batch", type=int, default=256, help="trajectory windows per GPU per optimizer step")
    args = ap.parse_args()
    select_workload(args.workload, args.batch)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # before anything in this process touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    force_dp = os.environ.get("MADIFF_FORCE_DP") == "1"      # rehearse the data-parallel step in a 1-rank RCCL group
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from madiff_amd import ops
    from madiff_amd.trainer import FlatAdamEMA, GraphedTrainStep, train_step
    ops.set_precision("bf16")
    torch.manual_seed(0)
    dif = build(device)
    ema = copy.deepcopy(dif)
    opt = FlatAdamEMA(dif, ema, lr=2e-4, ema_decay=0.995, bucket_mb=float(os.environ.get("MADIFF_BENCH_BUCKET_MB", "64")))
    opt.broadcast_parameters(0)
    batch = synthetic_batch(BATCH, device, seed=rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 2)):
        train_step(dif, opt, [batch], do_ema=1)
    # single process: the whole step (loss fwd, bwd, fused Adam/EMA) is one captured hipGraph, replayed per step.
    # data parallel: the graph holds forward + backward with one graph-external event per gradient bucket; the RCCL
    # all-reduces run eagerly on the communication stream behind those events (overlapping the replayed backward), then
    # the fused Adam.  If capturing fails the job falls back to the eager step (same arithmetic).
    use_graph = not args.no_graph
    graphed = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(dif, opt, batch)
        except Exception as e:                       # noqa: BLE001
            if world == 1 and not force_dp:
                raise
            print(f"[rank {rank}] graph capture of the data-parallel step failed ({e!r}); using the eager step", flush=True)
            use_graph = False
    if world > 1:                                    # every rank must take the same path
        flag = torch.tensor([1 if graphed is not None else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            graphed, use_graph = None, False

    def one_step(i):
        ema_mode = 1 if (i % 10 == 9) else 0
        if graphed is not None:
            return graphed.step(batch, do_ema=ema_mode)[0]
        return train_step(dif, opt, [batch], do_ema=ema_mode)[0]

    if graphed is not None:
        one_step(0)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # per-launch HIP-event timing of the GEMM kernels (roofline).  Graph path: a second capture of the same step with a
    # timing event-record node before and after every GEMM launch (on the launch's own stream), replayed K times; the
    # pairs are read after each replay.  So the durations are those of the real step (graph-paced launches, the step's
    # own concurrency between branches).  Eager path / fallback: the same K steps issued eagerly with a torch event
    # pair around every GEMM launch.
    summ, timed_on, dt_inst = None, "", 0.0
    if graphed is not None:
        try:
            gt = ops.GraphKernelTimer()
            ops.GRAPH_TIMER = gt
            try:
                graphed_t = GraphedTrainStep(dif, opt, batch)
            finally:
                ops.GRAPH_TIMER = None
            barrier()
            e0 = time.perf_counter()
            for i in range(args.steps):
                graphed_t.step(batch, do_ema=1 if (i % 10 == 9) else 0)
                gt.collect()
            barrier()
            dt_inst = time.perf_counter() - e0
            summ = gt.summary()
            gt.close()
            timed_on = (f"a second captured graph of the same step with a timing event-record node before and after every GEMM "
                        f"launch, replayed {args.steps} times right after the timed region "
                        f"({1e3 * dt_inst / args.steps:.2f} ms/step incl. reading {len(gt.totals)} kernel families' event pairs)")
            del graphed_t
        except Exception as e:                       # noqa: BLE001
            print(f"[rank {rank}] in-graph kernel timing failed ({e!r}); timing an eager pass instead", flush=True)
            ops.GRAPH_TIMER = None
            summ = None
    if summ is None:
        timer = ops.KernelTimer()
        ops.TIMER = timer
        barrier()
        e0 = time.perf_counter()
        for i in range(args.steps):
            train_step(dif, opt, [batch], do_ema=1 if (i % 10 == 9) else 0)
        barrier()
        dt_inst = time.perf_counter() - e0
        ops.TIMER = None
        summ = timer.summary()
        timed_on = ("an instrumented eager pass of the same K steps right after the timed region "
                    f"({1e3 * dt_inst / args.steps:.2f} ms/step with the event pairs)")
    train_sps = BATCH * world * args.steps / dt

    # ---- sampling (BASELINE.json configs[4] on the default run): the episode batch is sharded over the ranks
    # (madiff_amd.sharding.sample_sharded: contiguous slices, each rank replays its own captured K-step loop with the EMA
    # weights, no collective in the loop) and ONE final gather brings the plans to rank 0; the gather is inside the timed
    # region.  The sampling shape defaults to spread for the headline run (configs[4] names it) and to the training
    # workload otherwise (configs[3]: SMAC train + sample).
    sample = None
    if not args.no_sample:
        from madiff_amd.sharding import sample_sharded
        sw_name = args.sample_workload or ("spread" if args.workload == "tag" else args.workload)
        sb = args.sample_batch
        if sw_name != args.workload:
            select_workload(sw_name, args.batch)
            torch.manual_seed(1)
            sampler = build(device)            # random-init weights of that shape (synthetic, like the training leg)
            if world > 1:
                for p_ in sampler.parameters():
                    dist.broadcast(p_.data, src=0)
        else:
            sampler = ema
        sbatch = synthetic_batch(sb * world, device, seed=1000)          # the whole episode batch, as the evaluator holds it
        cond = {"x": sbatch["cond"]["x"].contiguous(), "masks": sbatch["cond"]["masks"].contiguous()}
        ret = sbatch["returns"].contiguous()
        sample_sharded(sampler, cond, ret, dst=0)                       # warm-up (allocator, pack cache, graph capture)
        barrier()
        s0 = time.perf_counter()
        xs = sample_sharded(sampler, cond, ret, dst=0)
        barrier()
        sdt = time.perf_counter() - s0
        if world > 1:
            tt = torch.tensor([sdt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sdt = float(tt.item())
        if rank == 0:
            assert xs.shape[0] == sb * world
            tps = sb * world / sdt
            tf = tps * K_DIFF * 2 * FWD_GF_PER_SAMPLE / 1e3 / world      # model TFLOP/s per GPU (2 forwards per step)
            sample = {"metric": "DDPM sampled trajectories/sec", "value": tps, "unit": "trajectories/s",
                      "workload": WORKLOAD_DESC, "episodes": sb * world, "episodes_per_gpu": sb, "k_steps": K_DIFF,
                      "guidance": "2B-batch CFG, w=1.2", "graph": SAMPLE_GRAPH_DESC,
                      "collectives": "none in the loop; one final gather to rank 0 (inside the timed region)",
                      "seconds": sdt, "finite": bool(torch.isfinite(xs).all().item()),
                      "inpainting_exact": bool(torch.equal(xs[:, : HH + 1], cond["x"][:, : HH + 1])),
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": tf / MFMA_BF16_PEAK_TF,
                                   "algorithmic_tflop_per_trajectory": K_DIFF * 2 * FWD_GF_PER_SAMPLE / 1e3}}
            if world == 1 and not args.no_cpu:
                sample["cpu_baseline"] = cpu_baseline_sample()
        if sw_name != args.workload:
            del sampler
            select_workload(args.workload, args.batch)

    if rank == 0:
        dom = max((k for k in summ if "gemm" in k or "conv_taps" in k), key=lambda k: summ[k]["ms"])   # dominant kernel
        d = dict(summ[dom])
        by_variant = None
        if dom.startswith("conv_taps_bf16_kernel"):
            # one kernel template, two timing labels: report the WHOLE family (forward + data gradient + the <GNB> data
            # gradients with the GroupNorm-backward prologue), which is also what the PMC traffic record aggregates
            dom = "conv_taps_bf16_kernel"
            parts = {k: v for k, v in summ.items() if k.startswith("conv_taps_bf16_kernel")}
            d = {f: sum(v.get(f, 0.0) for v in parts.values()) for f in ("flops", "ms", "launches")}
            by_variant = {k: {"launches_per_step": v["launches"] / args.steps, "avg_us": 1e3 * v["ms"] / v["launches"],
                              "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in parts.items()}
        traffic = load_traffic(dom, d["launches"] / args.steps)
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        kernels = {}
        for k, v in summ.items():
            rec = {"launches_per_step": v["launches"] / args.steps, "avg_us": 1e3 * v["ms"] / v["launches"],
                   "share_of_step": v["ms"] / (dt * 1e3)}
            if v.get("bytes", 0.0) > 0 and v.get("flops", 0.0) > 0:      # a fused launch: MFMA work + an HBM-bound epilogue
                rec.update({"bound": "mfma", "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                            "epilogue_algorithmic_mb_per_launch": v["bytes"] / v["launches"] / 1e6})
            elif v.get("bytes", 0.0) > 0:        # HBM-bound family: algorithmic bytes / measured time against 8 TB/s
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                rec.update({"bound": "hbm", "algorithmic_mb_per_launch": v["bytes"] / v["launches"] / 1e6, "gbps": gbs,
                            "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
            else:
                rec.update({"bound": "mfma", "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12})
            kernels[k] = rec
        fwd_flops_step = FWD_GF_PER_SAMPLE * 1e9 * BATCH
        out = {
            "metric": "denoiser train samples/sec", "value": train_sps, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC + f"; K={K_DIFF}, dim 128 x [1,4,8], batch {BATCH} per GPU",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "parallelism": f"dp{world}",
                       "step": "loss fwd + bwd + bucketed all-reduce + fused Adam (+EMA every 10)",
                       "launch": ("one hipGraph replay per step" if not opt.dp else
                                  "hipGraph replay of fwd+bwd, bucketed RCCL all-reduce behind graph-external events, eager "
                                  "fused Adam") if use_graph else "eager"},
            "loss": float(loss),
            "roofline": {"bound": "mfma", "kernel": KERNEL_DESC.get(dom, dom),
                         "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TF,
                         "traffic": traffic, "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                         "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                         "launches_per_step": d["launches"] / args.steps,
                         "by_variant": by_variant, "timed_on": timed_on},
            "kernels": kernels,
            "model_tflops_per_gpu": 3 * fwd_flops_step * args.steps / dt / 1e12,
            "sample": sample,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1:
            try:
                out["roofline"]["same_box_matmul"] = same_box_matmul(device, n_seq=BATCH * A, t=max(1, T // 4))
            except Exception as e:                   # noqa: BLE001
                print(f"bench.py: library comparison failed ({e!r})", file=sys.stderr, flush=True)
        if world == 1 and not force_dp and not args.no_dp1:
            out["extra"] = {"dp1_ms_per_step": dp1_ms_per_step(args)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
