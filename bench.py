#!/usr/bin/env python3
"""Headline benchmark of the DPoser hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE score-network training step (BASELINE.json metric M1) on a global batch of 65536
synthetic z-scored [B, 63] pose vectors: in-kernel perturbation (t ~ U, z ~ N), forward with dropout,
DSM loss, full backward into the flat fp32 gradient, RCCL all-reduce (N > 1), global-norm clip +
Adam + EMA.  The global batch is fixed as N grows ("B = 65536 @ 1/2/4/8 GPU") => strong scaling
(`--per-gpu-batch 65536` keeps the per-GPU batch instead and reports "scaling": "weak").
Rank 0 prints ONE JSON line; besides the contract's keys it carries
  roofline      -- live HIP-event timing of the dominant MFMA GEMM kernel over the timed region
  cpu_baseline  -- the CPU oracle (oracle/score_ref.py, torch-CPU port of the reference path) on the host cores
  extra         -- M2 (1000-step sub-VP sampling, samples/s), M3 (SMPL-X FK joints, poses/s) and full LBS (vertices) of this run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GLOBAL_BATCH = 65536
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md)
FALLBACK_KIND = "gn_fwd_train"      # (only if the untimed all-kinds pass below records nothing)
HBM_PEAK_GBS = 8000.0


_EPI_CLASS = {"gn_fwd": "EpiGN", "gn_fwd_train": "EpiGN<train>", "bias_silu": "EpiBiasSiLU<train>", "rowmajor": "EpiRowMajor",
              "plain": "EpiPlainFT", "gn_bwd_dgrad": "EpiGNBwd", "silu_bwd_dgrad": "EpiSiLUBwd", "wgrad": "EpiWgrad"}


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_hbm_traffic.json, written by tools/profile_bench.sh: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2
    read correction).  PMC counters cannot be collected from inside the process, so this is a lookup; None if absent."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
        head, epi = kernel.rstrip(">").rsplit(",", 1)
        row = table.get(f"{head},{_EPI_CLASS[epi]}>")
        return None if row is None else (row["read_MB"] + row["write_MB"]) * 1e6
    except Exception:
        return None


def pmc_mfma(kernel):
    """Matrix-pipe counters of ``kernel`` from the committed SQ / GRBM passes of this same command (profiles/pmc_mfma.json, written by
    tools/profile_bench.sh): MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) and the effective
    shader clock GRBM_GUI_ACTIVE / 8 / duration.  A lookup like ``pmc_traffic``; None if absent."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))
        head, epi = kernel.rstrip(">").rsplit(",", 1)
        row = table.get(f"{head},{_EPI_CLASS[epi]}>")
        return None if row is None else {"mfma_busy_frac": row["mfma_busy_frac"], "clock_ghz": row["clock_ghz"],
                                         "issue_stall_frac": row.get("issue_stall_frac"), "parked_frac": row.get("parked_frac")}
    except Exception:
        return None


def pmc_fk_bytes_per_pose():
    """HBM bytes per pose of the joints-only FK kernel from the committed PMC passes: the bench launches it at 2^20 (23 launches) and
    2^22 poses (12 launches); the table holds the per-launch average over all of them."""
    try:
        row = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))["k_fk_joints_dma<KinSMPLX>"]
        poses = (23 * (1 << 20) + 12 * (1 << 22)) / 35.0
        return (row["read_MB"] + row["write_MB"]) * 1e6 / poses if row["launches"] == 35 else None
    except Exception:
        return None


LBS_FWD_KERNELS = ("k_fk", "k_split_pf", "gemm_ft_kernel<bf16,256x256,EpiWgrad>", "k_skin", "k_extra_joints")


def pmc_lbs_bytes(with_backward):
    """HBM bytes of one LBS call from the committed PMC passes (profiles/pmc_lbs_traffic.json, written by tools/lbs_pmc.sh: per-kernel
    FETCH_SIZE / WRITE_SIZE of a 4096-pose call, summed); None if absent."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_lbs_traffic.json")))
        return table["fwd_bwd_bytes" if with_backward else "fwd_bytes"]
    except Exception:
        return None


def synthetic_poses(n, device, seed=42):
    """rows of the reference's examples/toy_data.npz (shipped as a fixture) sampled with replacement,
    z-scored with axis_normalize2."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g10_normalizer.npz"))
    poses = torch.tensor(g["toy_pose_samples"])
    mean, std = torch.tensor(g["stats/axis_normalize2/mean_poses"]), torch.tensor(g["stats/axis_normalize2/std_poses"])
    idx = torch.randint(0, poses.shape[0], (n,), generator=torch.Generator().manual_seed(seed))
    return ((poses[idx] - mean) / std).to(device), poses[idx].to(device)


def _host():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, model


def cpu_baseline(budget_s=10.0):
    """Reference-path CPU port (the oracle: torch-CPU restatement pinned to the reference's goldens) timed on this box's host cores,
    SURVEY 8d: the train step at B = 8192 (primary object) and, as ``legs``, the train step at the reference's default batch
    1280, the 1000-step sampler at B = 500 and SMPL-X LBS at B = 8192 -- each on a bounded sample (~10 s) of the workload."""
    from oracle import fk_ref
    from oracle import score_ref as R
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.configs import load_config
    avail, cpu_model = _host()
    cores = min(avail, 32)          # torch-CPU GEMMs at these sizes thrash on a 256-thread host (measured: 81 s/step at 256 threads)
    torch.set_num_threads(cores)
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    torch.manual_seed(42)
    m = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2)
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    p["sigmas"] = R.sigma_table()
    gen = torch.Generator().manual_seed(0)

    def timed(fn, budget):
        t0 = time.perf_counter()
        fn()                                                # warm-up (also sizes the sample)
        warm = time.perf_counter() - t0
        n_target = max(1, min(30, int(budget / max(warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n_target):
            fn()
        return n_target, time.perf_counter() - t0

    def train_leg(B, budget):
        st = R.TrainState({k: v.clone() for k, v in p.items()}, R.param_names())
        batch, _ = synthetic_poses(B, "cpu")

        def one():
            t = torch.rand(B, generator=gen) * (1 - 1e-5) + 1e-5
            z = torch.randn(B, 63, generator=gen)
            masks = [(torch.rand(B, 1024, generator=gen) >= 0.1).float() for _ in range(5)]
            R.train_step(st, R.SubVP(), batch, t, z, drop_masks=masks, drop_p=0.1)

        n, el = timed(one, budget)
        return {"value": B * n / el, "unit": "poses/s", "sample": f"{n} train steps at B={B}"}

    legs = {"train_b1280": train_leg(1280, budget_s * 0.6)}
    # sampler: B = 500, N = 1000 -- a bounded run of consecutive reverse steps, scaled to the 1000 of the metric
    x = torch.randn(500, 63, generator=gen)
    sde = R.SubVP(N=1000)
    ts = torch.linspace(1.0, 1e-3, 1000)
    state = {"x": x, "i": 0}

    def em():
        with torch.no_grad():
            i = state["i"] % 1000
            state["x"], _ = R.em_step(p, sde, state["x"], torch.ones(500) * ts[i], torch.randn(500, 63, generator=gen))
            state["i"] += 1

    n, el = timed(em, budget_s * 0.6)
    legs["sampler_b500_n1000"] = {"value": 500 / (el / n * 1000), "unit": "samples/s",
                                  "sample": f"{n} consecutive Euler-Maruyama steps at B=500, scaled to N=1000 steps per sample"}
    # SMPL-X LBS (vertices + 127 joints), numpy fp32 restatement of smplx lbs, on a slice of the B = 8192 workload
    asset = make_synthetic_smplx_asset(seed=0)
    _, raw = synthetic_poses(8192, "cpu")
    nb = 256
    n, el = timed(lambda: fk_ref.smplx_forward(asset, raw[:nb].numpy(), dtype=np.float32), budget_s * 0.4)
    legs["lbs_full_b8192"] = {"value": nb * n / el, "unit": "poses/s", "sample": f"{n} x {nb} poses of the B=8192 batch (numpy fp32 LBS, 10475 vertices)"}
    main_leg = train_leg(8192, budget_s)
    return {"value": main_leg["value"], "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": f"{main_leg['sample']} (BASELINE config 2 batch), torch-CPU oracle, {torch.get_num_threads()} threads",
            "cpu_model": cpu_model, "host_cpus": os.cpu_count(), "legs": legs,
            "threads_policy": f"min(host threads, 32) = {cores}: BASELINE.md section 4 says os.cpu_count() threads, but on this pool's 256-thread EPYC "
                              "hosts torch-CPU's GEMMs of these shapes collapse with every thread in use (measured in round 2: 81 s per B = 8192 "
                              "train step at 256 threads against ~1.9 s at 32) -- the 32-thread figure is the FASTER, i.e. fairer, CPU baseline"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH)
    ap.add_argument("--per-gpu-batch", type=int, default=0,
                    help="weak scaling: fix the batch per GPU (global batch = this x world size) instead of the global batch")
    ap.add_argument("--sampler-steps", type=int, default=1000)
    ap.add_argument("--no-extra", action="store_true", help="skip the sampler / FK measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-live-roofline", action="store_true",
                    help="no HIP events around the dominant GEMM kind inside the timed region (A/B timing of small batches)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start the N ranks ourselves -- as CHILD processes of a parent that has not touched the GPU
        # (counting devices does not initialise it), like the reference's multi-GPU entry spawns one process per GPU
        # (run/completion.py:326-338).  Rank 0 of the children prints the JSON line on the inherited stdout.
        import socket
        import subprocess
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get("DPOSER_BENCH_ALLOW_SHARED_GPU") != "1":
            print(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible: refusing to report a {args.gpus}-GPU number from fewer devices "
                  "(DPOSER_BENCH_ALLOW_SHARED_GPU=1 lets test rigs stack ranks on one device over gloo)", file=sys.stderr)
            sys.exit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    from dposer_amd import _C
    from dposer_amd import distributed as ddp
    from dposer_amd.algorithms.advanced import losses, sampling, sde_lib
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    from dposer_amd.configs import load_config

    rank, world, local_rank = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)     # (test rigs may run several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.per_gpu_batch > 0:
        args.global_batch = args.per_gpu_batch * world
    B_local = args.global_batch // world

    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    cfg.seed = 42
    torch.manual_seed(42)                                   # run/train.py:414
    model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=cfg.model.HIDDEN_DIM, embed_dim=cfg.model.EMBED_DIM,
                         n_blocks=cfg.model.N_BLOCKS)
    model.precision = args.precision
    model.to(dev)
    model._rng_seed = 42 + rank                             # per-rank Philox streams
    ddp.broadcast_(model.flat_params())
    sde = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
    state = dict(optimizer=losses.get_optimizer(cfg, model.parameters()), model=model,
                 ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=0)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    lo, hi = ddp.shard_bounds(args.global_batch, world, rank)
    batch_all, raw_all = synthetic_poses(args.global_batch, "cpu")
    batch = batch_all[lo:hi].to(dev).contiguous()

    for _ in range(args.warmup):
        step_fn(state, batch)
    # which GEMM kind dominates the step?  Decided HERE, by an untimed pass with events around every GEMM launch (two steps), not by a
    # constant from an old profile: the kind with the largest total time is the one the timed region brackets
    torch.cuda.synchronize()
    _C.profile_enable(True)
    for _ in range(2):
        step_fn(state, batch)
    torch.cuda.synchronize()
    pre = _C.profile_collect()
    _C.profile_enable(False)
    dominant_kind = FALLBACK_KIND
    if pre:
        dominant_name = max(pre.items(), key=lambda kv: kv[1][0])[0]
        for kname in _C.PROFILE_EPI_KINDS:
            if dominant_name.rstrip(">").endswith("," + kname):
                dominant_kind = kname
    ddp.barrier()
    torch.cuda.synchronize()
    # live roofline: HIP events around the launches of the dominant GEMM kind only (bracketing all ~30 GEMM launches of a
    # step costs ~3 % of it; the full per-kind table comes from three extra, untimed steps below)
    if ddp.dp_active():
        ddp.stats_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        # ... and only in every 4th step: the two event records per launch keep dependent kernels ~10 us apart
        if args.no_live_roofline:
            pass
        elif i % 4 == 0:
            _C.profile_enable(True, only=dominant_kind)
        elif i % 4 == 1:
            _C.profile_pause()
        out = step_fn(state, batch)
    torch.cuda.synchronize()
    ddp.barrier()
    elapsed = time.perf_counter() - t0
    dp_stats = ddp.stats_collect() if ddp.dp_active() else None
    ddp.stats_enable(False)
    prof = _C.profile_collect()
    _C.profile_enable(False)
    _C.profile_enable(True)
    for _ in range(3):
        step_fn(state, batch)
    torch.cuda.synchronize()
    prof_all = _C.profile_collect()
    _C.profile_enable(False)
    loss = float(out["step_loss"])
    own_elapsed = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt[0])
    ms_per_step = elapsed / args.steps * 1e3
    value = args.global_batch * args.steps / elapsed

    # ---- live roofline of the dominant GEMM kernel (HIP events on the launch stream, timed region only) ----
    roofline = None
    kernels = {}
    if prof:
        tot_ms = sum(v[0] for v in prof_all.values())
        for name, (ms, cnt, fl) in sorted(prof_all.items(), key=lambda kv: -kv[1][0]):      # untimed pass, all kinds
            kernels[name] = {"launches": int(cnt), "avg_us": ms / cnt * 1e3, "share_of_gemm_time": ms / tot_ms,
                             "tflops": (fl / (ms * 1e-3)) / 1e12 if ms > 0 else None}
        name, (ms, cnt, fl) = max(prof.items(), key=lambda kv: kv[1][0])                     # timed region, dominant kind
        ach = (fl / (ms * 1e-3)) / 1e12
        roofline = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(name),
                    "traffic_source": "lookup: profiles/pmc_hbm_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command, tools/profile_bench.sh); NOT measured in this run", "launches": int(cnt), "avg_launch_us": ms / cnt * 1e3,
                    "flops_per_launch": fl / cnt, "gemm_time_share_of_step": tot_ms / 3 / ms_per_step}
        # the big kernel furthest below the roofline (>= 5 % of the GEMM time of the step), from the untimed all-kinds pass
        big = {k: v for k, v in kernels.items() if v["share_of_gemm_time"] >= 0.05 and v["tflops"]}
        if big:
            wname = min(big, key=lambda k: big[k]["tflops"])
            roofline["worst"] = {"kernel": wname, "achieved": big[wname]["tflops"], "frac": big[wname]["tflops"] / MFMA_BF16_PEAK_TFLOPS,
                                 "avg_launch_us": big[wname]["avg_us"], "launches": big[wname]["launches"],
                                 "share_of_gemm_time": big[wname]["share_of_gemm_time"], "measured": "HIP events around every GEMM launch of 3 untimed steps"}
            wm = pmc_mfma(wname)
            if wm:
                roofline["worst"].update(wm)
        roofline["kernel_chosen_by"] = "largest total GEMM time in an untimed all-kinds pass of this run (2 steps, HIP events around every launch)"
        roofline["peak_note"] = ("peak = the guide's dense bf16 figure at 2.4 GHz; under MFMA load this chip clocks 1.3-2.1 GHz (`clock_ghz`), and the K loop "
                                 "of this kernel issues an MFMA every 33 cycles of a SIMD (the pipe's floor): profiles/r05_tile_phase_probe.txt")
        mf = pmc_mfma(name)
        if mf:
            roofline.update(mf)
            roofline["mfma_counters_source"] = ("lookup: profiles/pmc_mfma.json (committed rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE "
                                                "passes of this command); NOT measured in this run")

    dropped = state["optimizer"].nonfinite_steps()
    if not np.isfinite(loss) or dropped:
        raise RuntimeError(f"training diverged inside the benchmark: last loss {loss}, {dropped} step(s) dropped for a non-finite gradient")
    dp_extra = None
    if dp_stats:
        # attribution of the data-parallel step on THIS rank (rank 0 prints): what the step waited for the collectives vs everything else;
        # the slowest rank's step time is `ms_per_step` (max over ranks), this rank's own wall time per step is `rank_ms_per_step`
        own_ms = own_elapsed / args.steps * 1e3
        eng = model._engine()
        dp_extra = dict(dp_stats, rank_ms_per_step=own_ms, rank_compute_ms_per_step=own_ms - dp_stats["exposed_allreduce_ms_per_step"],
                        gradient_buckets=len(eng.grad_buckets), flat_gradient_bytes=int(model.flat_params().numel()) * 4,
                        backend=torch.distributed.get_backend(), world=world,
                        note="bucket collectives are enqueued from inside the backward call as their ranges become final "
                             "(dposer_dsm_loss_fwd_bwd_notify -> StreamedAllReduce); the dead pre_dense_cond range belongs to no bucket")
    extra = {"train_loss_last_step": loss, "nonfinite_gradient_steps_dropped": dropped, "train_tflops_algorithmic": 42.59e6 * value / 1e12,
             "gemm_kernels": kernels}
    if dp_extra:
        extra["dp"] = dp_extra
    per_rank = None
    if ddp.is_initialized():
        # ---- self-validation of a multi-rank run (the first 8-GPU run has nobody to debug it): what the communicator says about itself,
        # which device every rank really ran on, every rank's own step time, and the N = 1 step of THIS build on every rank's GPU (global
        # batch, no collective: comparable with the single-GPU BENCH line; a slow GPU or a rank that shares a device shows up here)
        backend = torch.distributed.get_backend()
        comm = {"backend": backend, "world_size_reported_by_the_process_group": torch.distributed.get_world_size(), "launcher_world_size": world,
                "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None,
                "forced_one_rank_collectives": os.environ.get("DPOSER_DIST_FORCE_COLLECTIVES") == "1"}
        n1 = None
        if not args.no_extra or os.environ.get("DPOSER_BENCH_N1_LEG") == "1":
            opt = state["optimizer"]
            opt._ensure_flat()
            keep = [t.clone() for t in (model.flat_params(), opt._flat_m, opt._flat_v, state["ema"].flat_shadow_for(model.flat_params()))]
            keep_step, keep_count = state["step"], opt._step_count
            full = batch_all.to(dev).contiguous()
            with ddp.local_only():
                for _ in range(2):
                    step_fn(state, full)
                torch.cuda.synchronize()
                tn = time.perf_counter()
                n_n1 = max(3, min(args.steps, 10))
                for _ in range(n_n1):
                    step_fn(state, full)
                torch.cuda.synchronize()
                n1_ms = (time.perf_counter() - tn) / n_n1 * 1e3
            for dst, src in zip((model.flat_params(), opt._flat_m, opt._flat_v, state["ema"].flat_shadow_for(model.flat_params())), keep):
                dst.copy_(src)                                     # the replicas stepped on their own inside the block: back to the common state
            _C.bump_param_epoch()
            state["step"], opt._step_count = keep_step, keep_count
            del full, keep
            n1 = {"ms_per_step": n1_ms, "poses_per_s": args.global_batch / (n1_ms * 1e-3), "steps": n_n1, "global_batch": args.global_batch}
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "device_name": torch.cuda.get_device_name(dev),
                "device_uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")), "host": __import__("socket").gethostname(),
                "per_gpu_batch": hi - lo, "own_ms_per_step": own_elapsed / args.steps * 1e3, "n1_same_build": n1}
        per_rank = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(per_rank, mine)
        devices = [(r["host"], r["device_uuid"] or r["device_index"]) for r in per_rank]
        extra["ranks"] = {"communicator": comm, "per_rank": per_rank, "distinct_devices": len(set(devices)),
                          "all_ranks_on_distinct_devices": len(set(devices)) == len(devices),
                          "n1_same_build": {"what": "the single-process step (no collective) of this build at the GLOBAL batch, timed on every rank's own GPU",
                                            "ms_per_step_per_rank": [r["n1_same_build"]["ms_per_step"] if r["n1_same_build"] else None for r in per_rank],
                                            "poses_per_s_rank0": per_rank[0]["n1_same_build"]["poses_per_s"] if per_rank[0]["n1_same_build"] else None}}
    if not args.no_extra and args.precision != "fp32":
        # the same step in fp32 parity mode (exact-fp32 MFMA, 1/16 of the bf16 matrix rate): the mode the tight parity numbers
        # of the test suite come from, next to the bf16 headline
        model.precision = "fp32"
        for _ in range(2):
            step_fn(state, batch)
        ddp.barrier()
        torch.cuda.synchronize()
        t32 = time.perf_counter()
        n32 = 5
        for _ in range(n32):
            step_fn(state, batch)
        torch.cuda.synchronize()
        ddp.barrier()
        e32 = time.perf_counter() - t32
        if world > 1:
            tt = torch.tensor([e32], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            e32 = float(tt[0])
        extra["train_step_fp32_mode"] = {"poses_per_s": args.global_batch * n32 / e32, "ms_per_step": e32 / n32 * 1e3, "steps": n32,
                                         "tflops_algorithmic": 42.59e6 * args.global_batch * n32 / e32 / 1e12, "fp32_mfma_peak_tflops": 157.3,
                                         "roofline": {"bound": "mfma", "achieved": 42.59e6 * args.global_batch * n32 / e32 / 1e12, "peak": 157.3,
                                                      "unit": "TFLOP/s", "frac": 42.59e6 * args.global_batch * n32 / e32 / 1e12 / 157.3, "traffic": None,
                                                      "scope": "whole step (42.59 MFLOP per pose over the step's wall time), exact-fp32 MFMA "
                                                               "(v_mfma_f32_32x32x2_f32) peak"}}
        model.precision = args.precision
        model._engines.pop("fp32", None)                    # release the fp32 engine's packed weights / workspaces
        torch.cuda.empty_cache()
        # ... and in bf16x3 mode: reference-precision results ON the bf16 matrix pipe (operands split into two bf16 terms, three products per
        # term, fp32 accumulation; fp32 epilogues) -- the same goldens at the same tolerances as the fp32 mode (tests/test_gpu_score.py)
        model.precision = "bf16x3"
        for _ in range(2):
            step_fn(state, batch)
        ddp.barrier()
        torch.cuda.synchronize()
        tx3 = time.perf_counter()
        nx3 = 10
        for _ in range(nx3):
            step_fn(state, batch)
        torch.cuda.synchronize()
        ddp.barrier()
        ex3 = time.perf_counter() - tx3
        if world > 1:
            tt = torch.tensor([ex3], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            ex3 = float(tt[0])
        alg = 42.59e6 * args.global_batch * nx3 / ex3 / 1e12
        # per-kernel rates of this mode from live HIP events (three more, untimed steps): the dominant GEMM kind's roofline.  A launch's FLOPs
        # are the ALGORITHMIC ones (2 M N K of the layer); the matrix pipe executes three bf16 products per term, `matrix_pipe_frac` says so
        _C.profile_enable(True)
        for _ in range(3):
            step_fn(state, batch)
        torch.cuda.synchronize()
        prof_x3 = _C.profile_collect()
        _C.profile_enable(False)
        kernels_x3, dom_x3 = {}, None
        if prof_x3:
            tot_x3 = sum(v[0] for v in prof_x3.values())
            for kname, (ms, cnt, fl) in sorted(prof_x3.items(), key=lambda kv: -kv[1][0]):
                kernels_x3[kname] = {"launches": int(cnt), "avg_us": ms / cnt * 1e3, "share_of_gemm_time": ms / tot_x3,
                                     "tflops": (fl / (ms * 1e-3)) / 1e12 if ms > 0 else None}
            kname, (ms, cnt, fl) = max(prof_x3.items(), key=lambda kv: kv[1][0])
            ach3 = (fl / (ms * 1e-3)) / 1e12
            dom_x3 = {"bound": "mfma", "kernel": kname, "achieved": ach3, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach3 / MFMA_BF16_PEAK_TFLOPS,
                      "traffic": None, "launches": int(cnt), "avg_launch_us": ms / cnt * 1e3, "flops_per_launch": fl / cnt,
                      "matrix_pipe_frac": 3 * ach3 / MFMA_BF16_PEAK_TFLOPS, "gemm_time_share_of_step": tot_x3 / 3 / (ex3 / nx3 * 1e3),
                      "measured": "HIP events around every GEMM launch of 3 untimed steps in this mode; algorithmic FLOPs per launch (one product per term)"}
        extra["train_step_bf16x3_mode"] = {"poses_per_s": args.global_batch * nx3 / ex3, "ms_per_step": ex3 / nx3 * 1e3, "steps": nx3, "tflops_algorithmic": alg,
                                           "roofline": {"bound": "mfma", "achieved": alg, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                        "frac": alg / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                                                        "matrix_pipe_tflops": 3 * alg, "matrix_pipe_frac": 3 * alg / MFMA_BF16_PEAK_TFLOPS,
                                                        "scope": "whole step; `achieved` counts the ALGORITHMIC 42.59 MFLOP per pose -- the matrix pipe executes three "
                                                                 "bf16 products per term (hi*hi + lo*hi + hi*lo), `matrix_pipe_*` is that machine work"},
                                           "dominant_kernel_roofline": dom_x3, "gemm_kernels": kernels_x3,
                                           "parity": "fp32-mode tolerances against the reference goldens (forward 2e-5, gradients 2e-4, sampler 1e-4)"}
        model.precision = args.precision
        model._engines.pop("bf16x3", None)
        torch.cuda.empty_cache()
    if not args.no_extra:
        # ---- M2: 1000-step Euler-Maruyama sampling of the local shard, no trajectory kept ----
        model.eval()
        sde_s = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=args.sampler_steps)
        fn = sampling.get_sampling_fn(cfg, sde_s, (B_local, 63), lambda v: v, 1e-3, device=dev)
        # x_T ~ N(0, I) is the sampler's input: resident in HBM before the timed region; without z the sampler draws
        # it like the reference does, from the CPU generator (sampling.py:446)
        z_T = torch.randn(B_local, 63, device=dev, generator=torch.Generator(device=dev).manual_seed(42 + rank))
        # one untimed run first: it allocates the sampler's workspace and packs the weights (after the parity-mode legs above released
        # their multi-GB workspaces with empty_cache(), the first allocation alone was 0.3 s of a 0.6 s run)
        fn(model, z=z_T, traj_stride=0)
        ddp.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, xs = fn(model, z=z_T, traj_stride=0)
        torch.cuda.synchronize()
        ddp.barrier()
        t_s = time.perf_counter() - t1
        # per-kernel rate of the sampler from a short separate run with events (not inside the timed one)
        sde_p = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=min(50, args.sampler_steps))
        fn_p = sampling.get_sampling_fn(cfg, sde_p, (B_local, 63), lambda v: v, 1e-3, device=dev)
        _C.profile_enable(True)
        fn_p(model, z=torch.randn(B_local, 63, device=dev), traj_stride=0)
        torch.cuda.synchronize()
        sprof = _C.profile_collect()
        _C.profile_enable(False)
        t_s_own = t_s
        if world > 1:
            tt = torch.tensor([t_s], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t_s = float(tt[0])
        sps = args.global_batch / t_s
        extra["sampler"] = {"samples_per_s": sps, "seconds": t_s, "steps": args.sampler_steps, "finite": bool(torch.isfinite(xs).all()),
                            "tflops_algorithmic": 8.647e6 * args.sampler_steps * sps / 1e12}
        if args.precision == "bf16" and world == 1:
            # the same sampler in bf16x3 mode (reference-precision samples on the bf16 pipe), a bounded run of consecutive steps scaled to N
            model.precision = "bf16x3"
            n_x3 = min(100, args.sampler_steps)
            sde_x = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=n_x3)
            fn_x = sampling.get_sampling_fn(cfg, sde_x, (B_local, 63), lambda v: v, 1e-3, device=dev)
            fn_x(model, z=z_T, traj_stride=0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            _, xs3 = fn_x(model, z=z_T, traj_stride=0)
            torch.cuda.synchronize()
            t_x = (time.perf_counter() - t2) / n_x3 * args.sampler_steps
            alg_s = 8.647e6 * args.sampler_steps * (args.global_batch / t_x) / 1e12
            extra["sampler_bf16x3_mode"] = {"samples_per_s": args.global_batch / t_x, "seconds_scaled_to_n_steps": t_x, "steps_timed": n_x3, "steps": args.sampler_steps,
                                            "finite": bool(torch.isfinite(xs3).all()), "tflops_algorithmic": alg_s,
                                            "roofline": {"bound": "mfma", "achieved": alg_s, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                         "frac": alg_s / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "matrix_pipe_frac": 3 * alg_s / MFMA_BF16_PEAK_TFLOPS}}
            model.precision = args.precision
            model._engines.pop("bf16x3", None)
            torch.cuda.empty_cache()
        if sprof:
            name, (ms, cnt, fl) = max(sprof.items(), key=lambda kv: kv[1][0])
            ach_s = (fl / (ms * 1e-3)) / 1e12
            extra["sampler"]["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach_s, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": ach_s / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(name),
                                            "traffic_source": "lookup: profiles/pmc_hbm_traffic.json; NOT measured in this run",
                                            "launches": int(cnt), "avg_launch_us": ms / cnt * 1e3, "flops_per_launch": fl / cnt,
                                            "measured": f"HIP events around every GEMM launch of a separate {min(50, args.sampler_steps)}-step run"}
            extra["sampler"]["whole_run_frac_of_mfma_peak"] = extra["sampler"]["tflops_algorithmic"] / MFMA_BF16_PEAK_TFLOPS
        # ---- M3: SMPL-X forward kinematics, joints only ([B,63] -> [B,22,3]), HBM-bound: 516 B / pose ----
        from dposer_amd.body_model.body_model import BodyModel
        from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
        bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(dev)
        nfk = 1 << 20
        pose = raw_all[torch.randint(0, raw_all.shape[0], (nfk,), generator=torch.Generator().manual_seed(1))].to(dev).contiguous()
        for _ in range(3):
            bm.fk_joints(pose)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fk_runs = []
        for _rep in range(5):                # (the first launches after an idle synchronize() see lower clocks: every run is reported)
            e0.record()
            for _ in range(20):
                bm.fk_joints(pose)
            e1.record()
            torch.cuda.synchronize()
            fk_runs.append(e0.elapsed_time(e1) * 1e-3 / 20)
        fk_s = sorted(fk_runs)[len(fk_runs) // 2]          # median of the runs is the headline; mean and best are reported beside it
        fk_gbs = 516.0 * nfk / fk_s / 1e9
        extra["fk_joints"] = {"poses_per_s_per_gpu": nfk / fk_s, "batch": nfk,
                              "poses_per_s_mean_of_runs": nfk / (sum(fk_runs) / len(fk_runs)), "poses_per_s_best_run": nfk / min(fk_runs),
                              "roofline": {"bound": "hbm", "kernel": "k_fk_joints_dma<KinSMPLX>", "achieved": fk_gbs, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": fk_gbs / HBM_PEAK_GBS,
                                           "traffic": None if pmc_fk_bytes_per_pose() is None else pmc_fk_bytes_per_pose() * nfk,
                                           "traffic_source": "lookup: profiles/pmc_hbm_traffic.json (bytes per pose x 2^20); NOT measured in this run",
                                           "algorithmic_bytes_per_pose": 516, "avg_launch_us": fk_s * 1e6,
                                           "measured": "HIP events on the launch stream around 20 launches of 2^20 poses; median of five runs (all in runs_us)",
                                           "frac_mean_of_runs": 516.0 * nfk / (sum(fk_runs) / len(fk_runs)) / 1e9 / HBM_PEAK_GBS,
                                           "frac_best_run": 516.0 * nfk / min(fk_runs) / 1e9 / HBM_PEAK_GBS,
                                           "runs_us": [round(x * 1e6, 2) for x in fk_runs]}}
        # the same kernel on a 4x larger batch (4 GiB of poses + joints in HBM): launch tails and the ragged last wave weigh less
        nfk4 = 1 << 22
        pose4 = pose.repeat(4, 1)
        for _ in range(2):
            bm.fk_joints(pose4)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            bm.fk_joints(pose4)
        e1.record()
        torch.cuda.synchronize()
        fk4_s = e0.elapsed_time(e1) * 1e-3 / 10
        extra["fk_joints"]["at_4M_poses"] = {"poses_per_s_per_gpu": nfk4 / fk4_s, "achieved_GBps": 516.0 * nfk4 / fk4_s / 1e9,
                                            "frac_of_hbm_peak": 516.0 * nfk4 / fk4_s / 1e9 / HBM_PEAK_GBS}
        del pose4
        # ---- M3b: full linear blend skinning ([B,63] -> 10475 vertices + 127 joints), forward and forward+backward ----
        nl = 4096
        pb = pose[:nl].clone().requires_grad_(True)

        # incoming gradients are given (contiguous, as a loss on the vertices produces them): the timed region is LBS forward + LBS
        # backward.  (Up to round 3 this leg called (v.sum() + Jtr.sum()).backward(): torch's reduction of 125 MB and the materialisation of
        # its stride-0 gradient -- 516 MB -- were 0.25 ms of the reported time.)
        gv = torch.ones(nl, 10475, 3, device=dev)
        gj = torch.ones(nl, 127, 3, device=dev)

        def lbs_fwd_bwd(grad):
            if grad:
                out = bm(pose_body=pb)
                torch.autograd.backward([out.v, out.Jtr], [gv[:, :out.v.shape[1]], gj[:, :out.Jtr.shape[1]]])
                pb.grad = None
            else:
                with torch.no_grad():
                    bm(pose_body=pb)

        for grad in (False, True):
            for _ in range(3):
                lbs_fwd_bwd(grad)
            torch.cuda.synchronize()
            # (five runs of ten, median reported, every run listed: the first calls after an idle synchronize() run at a lower clock -- with three
            #  runs the median itself was such a run every other time: 1.39 / 1.41 / 1.29 ms in one bench, 1.31 in a dedicated A/B on the same box)
            secs = []
            for _rep in range(5):
                e0.record()
                for _ in range(10):
                    lbs_fwd_bwd(grad)
                e1.record()
                torch.cuda.synchronize()
                secs.append(e0.elapsed_time(e1) * 1e-3 / 10)
            sec = sorted(secs)[len(secs) // 2]
            # HBM roofline of the leg: the vertices + joints a pose must produce (10475 x 3 + 127 x 3 floats = 127.2 KB) -- the forward's
            # algorithmic bytes; forward + backward also reads the incoming vertex / joint gradients of the same size
            alg = (10475 * 3 + 127 * 3) * 4 * (2 if grad else 1) + 63 * 4 * (2 if grad else 1)
            gbs = alg * nl / sec / 1e9
            key = "lbs_full_fwd_bwd" if grad else "lbs_full_fwd"
            extra[key] = {"poses_per_s_per_gpu": nl / sec, "batch": nl, "ms": sec * 1e3, "ms_mean_of_runs": sum(secs) / len(secs) * 1e3,
                          "ms_best_run": min(secs) * 1e3, "runs_ms": [round(x * 1e3, 4) for x in secs],
                          "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_pose": alg, "traffic": pmc_lbs_bytes(grad),
                                       "traffic_source": "lookup: profiles/pmc_hbm_traffic.json (sum over the LBS kernels of one call); NOT measured in this run",
                                       "note": "the leg is several kernels (FK, pose-blend GEMM on the bf16 matrix pipe, skinning): `achieved` is the "
                                               "algorithmic output bytes over the whole leg's time, median of the runs"}}
        # the same two legs on an asset whose skinning weights are spatially coherent, as a real SMPL-X template's are (the default
        # synthetic asset skins every vertex to four RANDOM joints: the worst case for the skinning kernels' per-joint gathers)
        bm_c = BodyModel(make_synthetic_smplx_asset(seed=0, coherent_skinning=True)).to(dev)
        for grad in (False, True):
            def leg():
                if grad:
                    out = bm_c(pose_body=pb)
                    torch.autograd.backward([out.v, out.Jtr], [gv[:, :out.v.shape[1]], gj[:, :out.Jtr.shape[1]]])
                    pb.grad = None
                else:
                    with torch.no_grad():
                        bm_c(pose_body=pb)
            for _ in range(3):
                leg()
            torch.cuda.synchronize()
            secs = []
            for _rep in range(5):
                e0.record()
                for _ in range(10):
                    leg()
                e1.record()
                torch.cuda.synchronize()
                secs.append(e0.elapsed_time(e1) * 1e-3 / 10)
            sec = sorted(secs)[len(secs) // 2]
            extra[("lbs_full_fwd_bwd" if grad else "lbs_full_fwd") + "_coherent_skinning"] = {
                "poses_per_s_per_gpu": nl / sec, "batch": nl, "ms": sec * 1e3, "runs_ms": [round(x * 1e3, 4) for x in secs],
                "note": "synthetic asset with contiguous vertex ranges following one bone and its tree neighbours (make_synthetic_smplx_asset(coherent_skinning=True))"}
        del bm_c
        # the same forward + backward the way a loss on the outputs reaches it ((v.sum() + Jtr.sum()).backward(): torch's reduction and the
        # materialisation of its stride-0 gradient are inside the timed region) -- the measurement of rounds 1-2, kept comparable
        def lbs_loss_backward():
            out = bm(pose_body=pb)
            (out.v.sum() + out.Jtr.sum()).backward()
            pb.grad = None
        for _ in range(3):
            lbs_loss_backward()
        torch.cuda.synchronize()
        secs = []
        for _rep in range(3):
            e0.record()
            for _ in range(10):
                lbs_loss_backward()
            e1.record()
            torch.cuda.synchronize()
            secs.append(e0.elapsed_time(e1) * 1e-3 / 10)
        extra["lbs_full_fwd_bwd_through_loss_backward"] = {"ms_mean_of_runs": sum(secs) / len(secs) * 1e3, "runs_ms": [round(x * 1e3, 4) for x in secs],
                                                           "poses_per_s_per_gpu": nl / (sum(secs) / len(secs)), "batch": nl}

    if not args.no_extra and world == 1 and args.precision == "bf16":
        # ---- the latency-bound end of the same path (not the headline): the training step at the reference's own batch (1280 poses,
        # configs/default_amass_configs.py:22) and at the per-rank batch of the headline's 8-GPU leg (8192), BASELINE config 3 (1000-step sampler,
        # 500 poses).  Same model / optimizer state / kernels, other tilings (128x32 / 128x64 / 128x128); HBM-resident inputs, whole-call wall time.
        small = {}
        model.train()
        for nb, nsteps in ((1280, 200), (8192, 100)):
            xb = synthetic_poses(nb, dev, seed=7)[0]
            for _ in range(10):
                step_fn(state, xb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nsteps):
                step_fn(state, xb)
            torch.cuda.synchronize()
            e = time.perf_counter() - t1
            small[f"train_step_{nb}"] = {"ms_per_step": e / nsteps * 1e3, "poses_per_s": nb * nsteps / e, "steps": nsteps,
                                         "tflops_algorithmic": 42.59e6 * nb * nsteps / e / 1e12,
                                         "frac_of_mfma_peak": 42.59e6 * nb * nsteps / e / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        model.eval()
        sde_c3 = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=1000)
        fn_c3 = sampling.get_sampling_fn(cfg, sde_c3, (500, 63), lambda v: v, 1e-3, device=dev)
        z3 = torch.randn(500, 63, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        fn_c3(model, z=z3, traj_stride=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, x3 = fn_c3(model, z=z3, traj_stride=0)
        torch.cuda.synchronize()
        e = time.perf_counter() - t1
        small["sampler_500x1000"] = {"seconds": e, "samples_per_s": 500 / e, "us_per_step": e / 1000 * 1e6, "finite": bool(torch.isfinite(x3).all())}
        extra["small_batches"] = small
    if ddp.is_initialized() and not args.no_extra:
        # the legs that shard with no collective, as every rank measured them on its own shard / GPU
        legs = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(legs, {"rank": rank, "sampler_seconds": t_s_own, "sampler_samples": B_local,
                                                   "fk_poses_per_s": extra["fk_joints"]["poses_per_s_per_gpu"], "lbs_fwd_ms": extra["lbs_full_fwd"]["ms"],
                                                   "lbs_fwd_bwd_ms": extra["lbs_full_fwd_bwd"]["ms"]})
        extra["ranks"]["unsharded_legs_per_rank"] = legs
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        print(json.dumps({
            "metric": "poses/sec score-net train step (subVP DSM, ScoreModelFC) at global B=65536",
            "value": value, "unit": "poses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.per_gpu_batch > 0 else "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "subVP denoising-score-matching train step (fwd+bwd+clip+Adam+EMA), ScoreModelFC H=1024 E=512 2 blocks, "
                                   "z-scored toy-pose rows [B,63]", "global_batch": args.global_batch, "per_gpu_batch": B_local,
                       "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra}))
    if ddp.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    from dposer_amd.distributed import run_fail_fast
    run_fail_fast(main)
