#!/usr/bin/env python3
"""Headline benchmark: glimpse-patches/sec of the JoliNeedle rollout hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2]/[3], the configuration the metric is quoted on):
gpt-nano + yolox-nano patch encoder, 448 px patches, seq-len 20, --enable-stop, batch 64
agents per GPU on synthetic 4480x4480 fp32 images with 1-3 random boxes, one rank per GPU
(weak scaling: configs[3] is 512 = 8 x 64).  One "step" = one REINFORCE iteration
(src/reinforce.py:302-353): env build + whole-batch rollout of 20 glimpse steps with train-mode
BatchNorm (forced non-STOP actions so that S == T: a fixed amount of work) +
REINFORCE loss + backward through all 20 steps + ONE flat-gradient all-reduce (N > 1) +
clip_grad_value_ + AdamW.  `--mode rollout` times the forward-only (eval) rollout instead.
Inputs are resident in HBM before timing.

Prints ONE JSON line on rank 0 (schema in the task contract) with `roofline` for the nano
PAFPN conv stack (HIP events around the conv section of every glimpse step, on the stream
the kernels run on) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# SURVEY.md §8(d) / BASELINE.md §2: conv activation traffic of the nano PAFPN @448, layer-wise,
# BN+SiLU fused, concat/upsample/focus free: 17.44 M elements per patch, fp32 storage.
NANO_448_ELEMS_PER_PATCH = 17.44e6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
# --config c5 (BASELINE configs[4] topology on one GPU): yolox-s PAFPN @640 = 8.113 GMAC / patch, dense
# 3x3 convs on fp32 MFMA -> 69 FLOP per fp32 byte, right of the fp32 ridge: priced against the dense fp32 MFMA peak.
S_640_GMAC_PER_PATCH = 8.113e9
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
# HBM traffic of ONE forward conv-stack pass at B=64, 448 px from the PMC counters (separate --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM, tools/pmc_traffic.py):
# profiles/r01_g_pmc_conv_stack_traffic_{train,eval}.txt — the eval pass fuses DWConv / shortcut adds and moves
# fewer bytes than the layer-wise algorithmic figure.
PMC_TRAFFIC_BYTES_B64_448 = {"train": 5.651e9, "rollout": 4.399e9}       # round-1 values, used when no newer file exists
# Backward of the same stack, ALGORITHMIC bytes: per conv layer read g_out and z_out, read x_in, write g_in
# = 2 (in + out) = 2 * 17.44 M elements per patch = 139.5 MB fp32.  The implementation of a layer whose BatchNorm-backward
# sums are not fused into a neighbour's kernel reads g_out and z_out a second time (+ 2 out = 2 * 7.35 M elements): that
# re-read is NOT algorithmic (36 of the 77 layers already avoid it) and is only reported as a labelled second figure.
NANO_448_BWD_ELEMS_PER_PATCH = 2 * 17.44e6
NANO_448_BWD_REREAD_ELEMS_PER_PATCH = 2 * 7.35e6


def pmc_traffic(mode):
    """(HBM bytes per glimpse step at B=64, 448 px, fp32; the profile file they come from) for `mode` = "train" / "rollout"
    (one forward conv-stack pass) or "backward" (the conv-stack backward of one glimpse step), from the committed PMC passes
    of THIS bench command — separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs summarised by
    tools/pmc_traffic.py into profiles/pmc_traffic_latest.json (rocprofv3 cannot run inside the timed process)."""
    f = ROOT / "profiles" / "pmc_traffic_latest.json"
    if f.exists():
        try:
            d = json.loads(f.read_text())
            return float(d[mode]), d.get("files", {}).get(mode, "profiles/pmc_traffic_latest.json")
        except (KeyError, ValueError):
            pass
    if mode in PMC_TRAFFIC_BYTES_B64_448:
        return PMC_TRAFFIC_BYTES_B64_448[mode], f"profiles/r01_g_pmc_conv_stack_traffic_{'train' if mode == 'train' else 'eval'}.txt"
    return None, None


def synth_inputs(B, G, P, seed, device):
    gen = torch.Generator(device=device).manual_seed(seed)
    images = torch.rand((B, 3, G * P, G * P), device=device, generator=gen)
    g = torch.Generator().manual_seed(seed)
    bboxes = torch.zeros((B, 3, 4), dtype=torch.long)
    for b in range(B):
        for k in range(int(torch.randint(1, 4, (1,), generator=g))):
            w, h = (int(torch.randint(32, P, (1,), generator=g)) for _ in range(2))
            x = int(torch.randint(0, G * P - w, (1,), generator=g))
            y = int(torch.randint(0, G * P - h, (1,), generator=g))
            bboxes[b, k] = torch.tensor([x, y, x + w, y + h])
    start = torch.randint(0, G, (B, 2), generator=g)
    return images, bboxes, start


def bench_detector(args, ja, model_config, dev, rank, world, dist):
    """Secondary workload: NeedleYOLOX loss branch + backward + optim_yolox step on B patches."""
    import ctypes as C
    from jolineedle_amd import _lib
    B, P = args.batch, args.patch_size
    model = ja.GPT(model_config(patch_size=P, block_size=4, image_processor="yolox-s"), max_batch=B, device=str(dev))
    model.sync_weights()
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    patches = torch.rand((B, 3, P, P), device=dev, generator=gen)
    g = torch.Generator().manual_seed(7 + rank)
    tg = torch.zeros((B, 3, 5))
    for b in range(B):
        for k in range(int(torch.randint(0, 3, (1,), generator=g))):
            w, h = (int(torch.randint(24, P // 2, (1,), generator=g)) for _ in range(2))
            x, y = int(torch.randint(0, P - w, (1,), generator=g)), int(torch.randint(0, P - h, (1,), generator=g))
            tg[b, k] = torch.tensor([0.0, x, y, x + w, y + h])
    tg = tg.to(dev)
    eng = model.engine()

    def step():
        model.engine_zero_grad()
        losses = model.yolox.loss_and_backward(patches, tg)
        _lib.check(eng.lib.jn_optimizer_step_group(eng.handle, 1, 1e-4, 0.01, 1.0, 1.0, _lib.current_stream(dev)), "opt")
        return losses

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        # yolox-s PAFPN + head @448: 3.976 + 2.521 GMAC forward per patch; training ~ 3x forward
        tflops = 3.0 * 2.0 * (3.976e9 + 2.521e9) * (P / 448.0) ** 2 * B * args.steps / dt / 1e12
        print(json.dumps({
            "metric": f"detector-training patches/sec ({P}px, yolox-s, SimOTA loss + backward + AdamW)",
            "value": round(B * args.steps / dt, 1), "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"secondary: NeedleYOLOX.forward(patches, targets) loss branch on {B} patches/GPU "
                                   f"of {P}px, yolox-s detector, 0-2 boxes per patch", "global_batch": B * world},
            "roofline": {"bound": "mfma", "kernel": "yolox-s PAFPN + head forward + backward (dense 3x3 / 1x1 on "
                                                    "v_mfma_f32_16x16x4_f32), ~3x the forward FLOPs",
                         "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None}}), flush=True)


def bench_supervised(args, ja, model_config, dev, rank, world, dist):
    """Secondary workload (BASELINE configs[0]/[1], SURVEY.md §8 a15 + §8f rank 3): one supervised iteration = teacher
    walks on the host (integers only) + two device gathers + teacher-forced step over B*T patches + detector step on the
    walks' detector patches + AdamW on both groups (src/supervised.py:844-902 without augmentation)."""
    B = 4 if args.batch == 64 else args.batch
    T = 8 if args.seq_len == 20 else args.seq_len
    P, G = args.patch_size, args.grid
    model = ja.GPT(model_config(patch_size=P, block_size=T, image_processor="yolox-s"), max_batch=B * T, device=str(dev))
    model.sync_weights()
    images, bboxes, _ = synth_inputs(B, G, P, 12345 + rank, dev)
    batch = {"image": images, "bboxes": bboxes, "class_id": torch.zeros(B, dtype=torch.long)}
    cfg = ja.CfgNode(patch_size=P, max_seq_len=T, min_keypoints=0, max_keypoints=2, binomial_keypoints=False,
                     stop_enabled=False, learning_rate=1e-4, yolo_lr=1e-4, gradient_accumulation=1, detection_enabled=True)
    trainer = ja.SupervisedTrainer(cfg, model)
    for i in range(args.warmup):
        trainer.train_iteration(batch, seed=i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    det_patches, host_s = 0, 0.0
    for i in range(args.steps):
        h0 = time.perf_counter()
        trainer.generate_trajectories(batch, seed=1000 + i)            # timed separately: host walk + gathers
        torch.cuda.synchronize()
        host_s += time.perf_counter() - h0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        m = trainer.train_iteration(batch, seed=1000 + i)
        det_patches += int(m["trajectories"]["patches_yolox"].shape[0])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({
            "metric": f"glimpse-patches/sec ({P}px, seq-len {T}) supervised step",
            "value": round(B * T * world * args.steps / dt, 1), "unit": "glimpse-patches/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[0]/[1]: supervised, gpt-nano + yolox-nano encoder + yolox-s detector, {P}px, "
                                   f"seq-len {T}, batch {B}/GPU, {G * P}x{G * P} synthetic images; teacher walks + device "
                                   f"gathers + teacher-forced step + detector step ({det_patches / args.steps:.1f} patches/iter) "
                                   f"+ AdamW x2", "global_batch": B * world, "seq_len": T,
                       "trajectory_generation_ms": round(host_s / args.steps * 1e3, 3)},
            "roofline": None}), flush=True)


def _pick_threads(oracle, P):
    """Thread count for the CPU oracle: a short sweep on the box (fwd + bwd of the patch encoder on 4 patches), because
    an oversubscribed MKL-DNN conv on a small batch is several times slower than the best count."""
    import os
    # CPUs this process may actually use: the affinity mask and the cgroup quota (a GPU box hands a 1-GPU job a 16-CPU
    # share of a much larger host; 128 threads on that share ran 10x slower than 16)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    cands = sorted({c for c in (4, 8, 16, 32) if c <= ncpu} | {min(ncpu, 32)})
    x = torch.rand(4, 3, P, P)
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = oracle.gpt_backbone(x)[-1]
            out.sum().backward()
            ts.append(time.perf_counter() - t0)
        if min(ts) < best_t:
            best, best_t = c, min(ts)
    oracle.zero_grad()
    torch.set_num_threads(best)
    return best, ncpu


def cpu_baseline(P, T, seed, train):
    """CPU oracle (pure PyTorch fp32 restatement, oracle/) on the host cores: a bounded sample of the same workload —
    B=4 agents x up to T glimpse steps, 1 warm-up + median of 3, thread count picked by a sweep; the
    number of steps is cut so that the whole leg takes ~30 s (a short probe run sets it).  4480x4480 images would need
    1 GB/agent on the host, so the sample uses a 3x3 grid (the patch content is what costs; the grid size does not)."""
    from oracle import env_ref, rollout_ref
    from oracle.gpt_ref import build_gpt_ref
    torch.manual_seed(seed)
    B, G = 4, 3
    oracle = build_gpt_ref(1, patch_size=P, block_size=T, with_detector=False, image_processor=None)
    oracle.train(train)
    threads, ncpu = _pick_threads(oracle, P)
    params = [p for n, p in oracle.named_parameters()]
    opt = torch.optim.AdamW(params, lr=1e-4)
    images = torch.rand(B, 3, G * P, G * P)
    bboxes = torch.tensor([[[10, 10, 200, 200]]] * B)
    start = torch.randint(0, G, (B, 2))

    def one(Tc, forced):
        env = env_ref.EnvRef(images, bboxes, P, Tc, 1, True)
        t0 = time.perf_counter()
        if train:
            opt.zero_grad()
            ro = rollout_ref.rollout(oracle, env, forced_actions=forced, start_positions=start)
            m = rollout_ref.reinforce_metrics(ro, 0.01, rollout_ref.ReturnNormaliser())
            m["loss"].backward()
            torch.nn.utils.clip_grad_value_(params, 1)
            opt.step()
        else:
            with torch.no_grad():
                ro = rollout_ref.rollout(oracle, env, forced_actions=forced, start_positions=start)
                rollout_ref.reinforce_metrics(ro, 0.01, rollout_ref.ReturnNormaliser())
        return time.perf_counter() - t0

    probe = one(2, torch.randint(0, 8, (B, 2)))            # also the warm-up of the allocator / oneDNN primitives
    probe = min(probe, one(2, torch.randint(0, 8, (B, 2))))
    rate = 2 * B / probe
    Tc = max(2, min(T, int(30.0 * rate / (4 * B))))
    forced = torch.randint(0, 8, (B, Tc))
    times = [one(Tc, forced) for _ in range(4)][1:]
    med = sorted(times)[len(times) // 2]
    what = "full REINFORCE iteration (fwd + bwd + clip + AdamW)" if train else "forward rollout + loss"
    return {"value": round(B * Tc / med, 2), "unit": "glimpse-patches/s", "cores": threads,
            "kind": "port", "sample": f"B={B} agents x T={Tc} steps (of {T}; cut to fit ~30 s), {P}px patches, 3x3-patch images, "
            f"forced actions, {what}, 1 warm-up + median of 3 ({med:.2f} s), {threads} torch threads picked by a sweep on a "
            f"{ncpu}-CPU host"}


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU; it starts one child per rank
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env, as torch.distributed.run would set them), relays their output
    and exits with the worst child code.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    code = 0
    for pr in procs:
        code = max(code, abs(pr.wait()))
    sys.exit(code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="agents per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--patch-size", type=int, default=448)
    ap.add_argument("--grid", type=int, default=10, help="image side in patches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["train", "rollout", "detector", "supervised"], default="train",
                    help="train (default, the headline): full REINFORCE iteration; rollout: inference rollout; detector: "
                         "secondary, one detector training step (yolox-s, SimOTA loss, backward, AdamW) on --batch patches")
    ap.add_argument("--config", choices=["c3", "c5"], default="c3",
                    help="c3 (default, the headline): gpt-nano + yolox-nano encoder, 448 px, T=20, B=64; "
                         "c5: gpt-mini + yolox-s encoder, 640 px, T=32, B=16 (BASELINE configs[4] topology, secondary)")
    ap.add_argument("--detect", action="store_true",
                    help="rollout mode only (secondary): run the yolox-s detector on every visited patch (do_detection)")
    ap.add_argument("--sample", action="store_true",
                    help="free-running trajectories (SURVEY.md §8d): sampled actions, episodes end at STOP / when every box is "
                         "found, the batch stops when every agent is done; reports the mean executed steps.  Default: forced "
                         "non-STOP actions, S = T (a fixed amount of work)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="activation storage / MFMA type; bf16 is the inference (rollout) mode only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args)
    arch = {}
    if args.config == "c5":
        arch = dict(model_type="gpt-mini", gpt_backbone="yolox-s")
        if args.batch == 64 and args.seq_len == 20 and args.patch_size == 448:
            args.batch, args.seq_len, args.patch_size = 16, 32, 640

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if os.environ.get("JN_BENCH_SAME_DEVICE"):                    # rehearsal: all ranks share GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    n_ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JN_BENCH_BACKEND", "nccl")      # "gloo" only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        n_ranks_seen = dist.get_world_size()

    import jolineedle_amd as ja
    from jolineedle_amd.config import model_config

    B, T, P, G = args.batch, args.seq_len, args.patch_size, args.grid
    torch.manual_seed(12345)
    if args.mode == "detector":
        return bench_detector(args, ja, model_config, dev, rank, world, dist)
    if args.mode == "supervised":
        return bench_supervised(args, ja, model_config, dev, rank, world, dist)
    assert not (args.dtype == "bf16" and args.mode == "train"), "bf16 is the inference mode: use --mode rollout"
    if args.detect:
        assert args.mode == "rollout", "--detect belongs to --mode rollout"
        arch = dict(arch, with_detector=True, image_processor="yolox-s")
    else:
        arch = dict(arch, with_detector=False, image_processor=None)
    model = ja.GPT(model_config(patch_size=P, block_size=T, act_dtype=args.dtype, **arch),
                   max_batch=B, device=f"cuda:{local_rank}")
    model.sync_weights()
    cfg = ja.CfgNode(max_seq_len=T, entropy_weight=0.01, stop_enabled=True, reward_norm=True, seed=12345 + rank,
                     learning_rate=1e-4, gradient_accumulation=1)
    trainer = ja.ReinforceTrainer(cfg, model)
    images, bboxes, start = synth_inputs(B, G, P, 12345 + rank, dev)
    forced = None if args.sample else torch.randint(0, 8, (B, T), generator=torch.Generator().manual_seed(777 + rank)).to(dev)
    eng = model.engine()
    eng.lib.jn_set_profiling(eng.handle, 1)

    train = args.mode == "train"

    def one_step():
        env = ja.NeedleGeneralEnv(images, bboxes, P, T, 1, True, engine=eng)
        if train:
            m = trainer.train_iteration(env, forced_actions=forced, start_positions=start, sample_actions=True,
                                        stop_early=not os.environ.get("JN_BENCH_NO_STOP_EARLY"))   # (measuring aid: no skip flags)
            return m["steps"], m["loss"]
        ro = trainer.rollout(env, forced_actions=forced, start_positions=start, keep_patches=False,
                             do_detection=args.detect, sample_actions=True)
        m = trainer.compute_metrics(ro)
        return ro["rewards"].shape[1], m["loss"]

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if dist is not None and train:
        trainer.allreduce_timing = []        # (start, end) CUDA events around the iteration's ONE gradient all-reduce

    import ctypes as C
    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    patches, conv_ms, bwd_ms, glimpse_steps = 0, 0.0, 0.0, 0
    for _ in range(args.steps):
        S, loss = one_step()
        patches += B * S
        glimpse_steps += S
        ms = C.c_float()
        eng.lib.jn_last_timing(eng.handle, 1, C.byref(ms))
        conv_ms += ms.value
        if train:
            eng.lib.jn_last_timing(eng.handle, 2, C.byref(ms))
            bwd_ms += ms.value
    torch.cuda.synchronize()
    own = time.perf_counter() - t0           # this rank's own K steps (before it waits for the others)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(patches)], device=dev, dtype=torch.float64)
    multi = None
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_patches = float(tmax[0]), float(tsum[1])
        # diagnostics of the multi-rank run: every rank's own time per step and its all-reduce time per step (CUDA events
        # on the launch stream: the time the stream spends in / waiting for the collective, i.e. RCCL itself plus the
        # wait for the slowest rank's backward); the timed steps only
        ar = trainer.allreduce_timing[-args.steps:] if train and getattr(trainer, "allreduce_timing", None) else []
        ar_ms = sum(a.elapsed_time(b) for a, b in ar) / max(len(ar), 1)
        mine = torch.tensor([own / args.steps * 1e3, ar_ms, conv_ms / max(glimpse_steps, 1), bwd_ms / max(glimpse_steps, 1)],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        cols = torch.stack(allr).cpu()       # [world, 4]
        r3 = lambda v: [round(float(x), 3) for x in v]
        multi = {"rank_ms_per_step": {"min": round(float(cols[:, 0].min()), 3), "max": round(float(cols[:, 0].max()), 3),
                                      "per_rank": r3(cols[:, 0])},
                 "allreduce_ms": {"mean": round(float(cols[:, 1].mean()), 4), "max": round(float(cols[:, 1].max()), 4),
                                  "per_rank": r3(cols[:, 1]), "bytes": int(model._optim_gpt_numel * 4) if train else 0,
                                  "what": "CUDA events on the launch stream around the ONE flat-gradient all-reduce of an "
                                          "iteration (collective + wait for the slowest rank); mean over the timed steps"},
                 "forward_ms_per_pass_per_rank": r3(cols[:, 2]), "backward_ms_per_step_per_rank": r3(cols[:, 3]),
                 "backend": os.environ.get("JN_BENCH_BACKEND", "nccl")}
    else:
        total_patches = float(patches)

    if rank == 0:
        conv_ms_per_launch = conv_ms / max(glimpse_steps, 1)  # one PAFPN pass over B patches
        headline_shape = (B, P, args.dtype) == (64, 448, "f32")
        esz = 2.0 if args.dtype == "bf16" else 4.0
        algo_bytes = NANO_448_ELEMS_PER_PATCH * (P / 448.0) ** 2 * esz * B
        achieved = algo_bytes / (conv_ms_per_launch * 1e-3) / 1e9
        actions = (f"sampled actions, early STOP (free-running: {glimpse_steps / args.steps:.2f} of {T} steps executed per "
                   f"trajectory batch)") if args.sample else "forced non-STOP actions (S=T)"
        out = {
            "metric": "glimpse-patches/sec (448px, seq-len 20) REINFORCE step",
            "value": round(total_patches / elapsed, 1), "unit": "glimpse-patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "n_ranks_seen": n_ranks_seen,
            "config": {"workload": f"configs[2]/[3]: REINFORCE rollout, gpt-nano + yolox-nano encoder, {P}px, "
                                   f"seq-len {T}, --enable-stop, {B} agents/GPU x {world} GPU, "
                                   f"{G * P}x{G * P} synthetic images, {actions}",
                       "global_batch": B * world, "seq_len": T,
                       "phase": ("full REINFORCE iteration: env build + rollout (train-mode BN) + loss + backward + "
                                 "flat-gradient all-reduce + clip + AdamW") if train else
                                "env build + whole-batch rollout (forward, eval-mode BN) + REINFORCE loss",
                       "parallelism": f"dp{world} (independent agents per rank; one RCCL all-reduce of the flat "
                                      f"gradient per iteration)"},
            "roofline": {"bound": "hbm", "kernel": "yolox-nano PAFPN forward conv stack (stem/dw3x3/pw_mfma/spp/upsample"
                                                   + ("/bn_finalize, train-mode BN" if train else "") +
                                                   "), one pass over the batch per glimpse step",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": headline_shape and pmc_traffic(args.mode)[0] or None,
                         "traffic_source": headline_shape and pmc_traffic(args.mode)[1] or None,
                         "ms_per_launch": round(conv_ms_per_launch, 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "timing": "HIP events around the conv section of every glimpse step, on the launch stream "
                                   "(jn_set_profiling / jn_last_timing); rocprofv3 kernel stats of the same command: "
                                   "profiles/r04_*_train_iteration_kernel_stats.csv"},
        }
        if train and args.config == "c3":
            # second entry: the step-batched conv-stack backward (embed_fpn + PAFPN), priced at its ALGORITHMIC traffic:
            # per conv layer read g_out and z_out, read x_in, write g_in = 2 (in + out) elements (139.5 MB / patch)
            bwd_bytes = NANO_448_BWD_ELEMS_PER_PATCH * (P / 448.0) ** 2 * 4.0 * B
            bwd_reread = NANO_448_BWD_REREAD_ELEMS_PER_PATCH * (P / 448.0) ** 2 * 4.0 * B
            bwd_ms_per_pass = bwd_ms / max(glimpse_steps, 1)
            bwd_ach = bwd_bytes / (bwd_ms_per_pass * 1e-3) / 1e9
            bwd_traffic, bwd_src = pmc_traffic("backward") if headline_shape else (None, None)
            out["roofline_backward"] = {
                "bound": "hbm", "kernel": "yolox-nano PAFPN conv-stack backward (step-batched: fused 1x1 / depthwise data + "
                                          "weight gradients, BN-backward reductions, stem weight gradient), per glimpse step",
                "achieved": round(bwd_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bwd_ach / HBM_PEAK_GBS, 4),
                "traffic": bwd_traffic, "traffic_source": bwd_src,
                "ms_per_launch": round(bwd_ms_per_pass, 4), "algorithmic_bytes_per_launch": int(bwd_bytes),
                "achieved_incl_one_bn_reduce_reread": round((bwd_bytes + bwd_reread) / (bwd_ms_per_pass * 1e-3) / 1e9, 1),
                "note": "algorithmic = 2 (in + out) elements per layer; the labelled second figure adds one re-read of g_out and "
                        "z_out per layer (what a separate BatchNorm-backward reduction pass costs) and is not a roofline claim",
                "timing": "HIP events around the conv-stack backward of the iteration (jn_last_timing(2)); per-op table: "
                          "profiles/r04_*_backward_table_f32.txt"}
            # third entry: the whole iteration against the same roof — forward + backward algorithmic bytes of every
            # executed glimpse step over the wall time of the iteration (everything else included in the time)
            it_bytes = (algo_bytes + bwd_bytes) * glimpse_steps
            it_ach = it_bytes / elapsed / 1e9
            out["roofline_iteration"] = {
                "bound": "hbm", "kernel": "whole REINFORCE iteration (conv-stack forward + backward algorithmic bytes of all "
                                          "glimpse steps over the iteration's wall time)",
                "achieved": round(it_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(it_ach / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_iteration": int(it_bytes / max(args.steps, 1))}
        if multi is not None:
            out["multi_rank"] = multi
        if args.detect:
            out["metric"] += " + yolox-s detection on every visited patch"
            out["config"]["phase"] += "; do_detection=True (yolox-s PAFPN + head + NMS per glimpse)"
            out["roofline"]["note"] = ("the encoder's conv stack is timed while the detector pass of the previous glimpse runs "
                                       "beside it on the engine's second stream: not comparable with the headline figure")
        if args.config == "c5":
            tflops = 2.0 * S_640_GMAC_PER_PATCH * (P / 640.0) ** 2 * B / (conv_ms_per_launch * 1e-3) / 1e12
            out["metric"] = f"glimpse-patches/sec ({P}px, seq-len {T}) REINFORCE step, gpt-mini + yolox-s encoder"
            out["config"]["workload"] = (f"configs[4] topology on {world} GPU: gpt-mini + yolox-s (dense 3x3) encoder, {P}px, "
                                         f"seq-len {T}, {B} agents/GPU, {actions}")
            out["roofline"] = {"bound": "mfma", "kernel": "yolox-s PAFPN forward conv stack (dense 3x3 + 1x1 on "
                                                          "v_mfma_f32_16x16x4_f32), one pass over the batch per glimpse step",
                               "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                               "ms_per_launch": round(conv_ms_per_launch, 4)}
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, T, 12345, train)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
