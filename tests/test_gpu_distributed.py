"""Data-parallel training step on the GPU: gradient buckets + overlapped all-reduce (two ranks sharing cuda:0 over gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_grad_buckets_partition_the_flat_buffer():
    """L + 1 buckets in backward completion order: layers 4..1 (the first with post_dense), layer 0's weights ("front A"), then layer
    0's GroupNorm affine + the shared time embedding ("front B").  Together they cover every parameter that gets a gradient, once;
    the dead pre_dense_cond range lies between front A and front B and belongs to no bucket (it is not all-reduced)."""
    from gpu_common import make_model
    cfg, m, p = make_model(3, precision="fp32", dropout=0.0)
    eng = m._engine()
    b = eng.grad_buckets
    assert len(b) == 6
    assert b[0][1] == eng.num_params and b[4][0] == 0
    for (lo, hi), (lo2, hi2) in zip(b[:4], b[1:4]):
        assert lo < hi and hi2 == lo                     # descending, contiguous, disjoint
    names = [n for n, _ in m.named_parameters()]
    off = dict(zip(names, m._offsets))
    assert b[0][0] == off["b2_dense2.weight"] and b[1][0] == off["b2_dense1.weight"]
    assert b[3][0] == off["b1_dense1.weight"] == b[5][1]
    assert b[4] == (0, off["pre_dense_cond.weight"])
    assert b[5][0] == off["pre_gnorm.weight"]
    covered = torch.zeros(eng.num_params, dtype=torch.int32)
    for lo, hi in b:
        covered[lo:hi] += 1
    dead = torch.zeros(eng.num_params, dtype=torch.bool)
    for lo, hi in eng.nograd:
        dead[lo:hi] = True
    assert int(covered.max()) == 1
    assert bool((covered[~dead] == 1).all())             # every live parameter in exactly one bucket
    lo_c, hi_c = off["pre_dense_cond.weight"], off["pre_gnorm.weight"]
    assert int(covered[lo_c:hi_c].sum()) == 0            # the dead range travels nowhere


def _steps(model_seed, batch, t, z, n_steps, world, rank, precision, likelihood_weighting=False):
    """n_steps optimisation steps on this rank's shard with injected draws; returns the flat parameters.
    ``likelihood_weighting`` takes the step off the fused path: loss.backward() through autograd, gradients in ``p.grad``."""
    from gpu_common import make_model
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    from dposer_amd import distributed as ddp
    cfg, m, p = make_model(model_seed, precision=precision, dropout=0.0)
    cfg.optim.warmup = 0                                   # full learning rate from step 0 (the shipped warm-up starts at lr = 0)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    opt = losses.get_optimizer(cfg, m.parameters())
    ema = ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate)
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True,
                                 likelihood_weighting=likelihood_weighting)
    state = dict(model=m, optimizer=opt, ema=ema, step=0)
    lo, hi = ddp.shard_bounds(batch.shape[1], world, rank)
    out = []
    for i in range(n_steps):
        r = step_fn(state, batch[i, lo:hi].cuda(), t=t[i, lo:hi].cuda(), z=z[i, lo:hi].cuda())
        out.append(float(r["step_loss"]))
    torch.cuda.synchronize()
    return m.flat_params().detach().cpu().clone(), opt._flat_m.detach().cpu().clone(), out


def _worker(rank, world, port, q, model_seed, batch, t, z, n_steps, precision="fp32", likelihood_weighting=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DPOSER_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dposer_amd import distributed as ddp
    import torch.distributed as dist
    ddp.init_from_env()
    torch.cuda.set_device(0)
    flat, mom, losses_ = _steps(model_seed, batch, t, z, n_steps, world, rank, precision, likelihood_weighting)
    q.put((rank, flat.numpy(), mom.numpy(), losses_))
    ddp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("precision,batched", [("fp32", None), ("bf16", "0"), ("bf16", "1"), ("bf16", "layer-lanes")])
def test_two_rank_bucketed_step_equals_single_process_step(precision, batched, tuning_env):
    """Mean-reduced DSM loss over equal shards: the averaged shard gradients are the full-batch gradient, so two ranks
    (bucketed all-reduce overlapped with the backward pass) must track the single-process run on the whole batch.
    bf16: with two split-K weight-gradient launches per layer (DPOSER_WGRAD_BATCHED=0), with the one-launch form forced (=1: every
    bucket event is recorded after the last reduction) and with one lane launch per layer (the default under data parallelism from 16384 samples per rank)."""
    if batched == "layer-lanes":
        tuning_env(DPOSER_WGRAD_LAYER_LANES="1")          # (default from 16384 samples per rank; forced for this small batch)
    elif batched is not None:
        tuning_env(DPOSER_WGRAD_BATCHED=batched)          # (spawned ranks inherit the environment)
    rs = np.random.RandomState(5)
    n_steps, B = 3, 256
    batch = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    t = torch.tensor(rs.uniform(1e-3, 1.0, (n_steps, B)).astype(np.float32))
    z = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    ref, ref_mom, ref_losses = _steps(9, batch, t, z, n_steps, 1, 0, precision)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 9, batch, t, z, n_steps, precision)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    f0, f1 = res[0][1], res[1][1]
    assert np.array_equal(f0, f1) and np.array_equal(res[0][2], res[1][2])    # ranks stay bit-identical
    # first moment = running mean of the (clipped, averaged) gradients: the direct check of the all-reduce
    m0, mr = res[0][2], ref_mom.numpy()
    # (bf16: the GroupNorm backward sums of a 128-sample shard and of the 256-sample batch round differently -> looser bounds)
    lo_prec = precision != "fp32"
    assert np.linalg.norm(m0 - mr) / np.linalg.norm(mr) < (5e-3 if lo_prec else 1e-4)
    # Adam normalises the update (|delta| <= lr per step), so the parameters get an absolute bound well below lr = 2e-4
    assert np.abs(f0 - ref.numpy()).max() < (1e-4 if lo_prec else 2e-5)
    mean_losses = [(a + b) / 2 for a, b in zip(res[0][3], res[1][3])]
    assert np.allclose(mean_losses, ref_losses, rtol=(2e-3 if lo_prec else 2e-5))


def _worker_rng(rank, world, port, q, x):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DPOSER_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dposer_amd import distributed as ddp
    import torch.distributed as dist
    from gpu_common import make_model
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    ddp.init_from_env()
    torch.cuda.set_device(0)
    cfg, m, p = make_model(3, precision="fp32", dropout=0.1)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    state = dict(model=m, optimizer=losses.get_optimizer(cfg, m.parameters()), ema=ExponentialMovingAverage(m.parameters(), 0.9999), step=0)
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    loss = float(step_fn(state, x.cuda())["step_loss"])          # in-kernel t, z, dropout: no injected draws
    q.put((rank, loss))
    ddp.barrier()
    dist.destroy_process_group()


def test_ranks_draw_different_noise():
    """Every rank must key its Philox streams differently (losses.py:110-111 draw fresh t, z per sample of the GLOBAL batch):
    two ranks fed the SAME shard get different losses; with a rank-independent key they would be bit-identical."""
    rs = np.random.RandomState(11)
    x = torch.tensor(rs.standard_normal((128, 63)).astype(np.float32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rng, args=(r, 2, port, q, x)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.isfinite(res[0]) and np.isfinite(res[1])
    assert abs(res[0] - res[1]) > 1e-3 * abs(res[0])


def _langevin_run(z0, noise, world, rank):
    from gpu_common import make_model
    from dposer_amd.algorithms.advanced import sampling, sde_lib
    from dposer_amd import distributed as ddp
    cfg, m, p = make_model(7, precision="fp32")
    cfg.sampling.corrector = "langevin"
    lo, hi = ddp.shard_bounds(z0.shape[0], world, rank)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)          # (Langevin's alpha = 1 - beta_i / N needs a fine grid to stay positive)

    class Args:
        task = "denoise"

    fn = sampling.get_sampling_fn(cfg, sde, (hi - lo, 63), lambda v: v, 1e-3, device="cuda:0")
    _, x = fn(m, z=z0[lo:hi].cuda(), noise=noise[:, :, lo:hi].cuda(), start_step=1000 - noise.shape[0], args=Args())
    torch.cuda.synchronize()
    return x.cpu()


def _worker_langevin(rank, world, port, q, z0, noise):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DPOSER_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dposer_amd import distributed as ddp
    import torch.distributed as dist
    ddp.init_from_env()
    torch.cuda.set_device(0)
    q.put((rank, _langevin_run(z0, noise, world, rank).numpy()))
    ddp.barrier()
    dist.destroy_process_group()


def test_langevin_step_size_uses_global_batch_means_under_data_parallelism():
    """sampling.py:296-298 takes the norm means over the WHOLE batch: two ranks with (ragged) shards must reproduce the
    single-process samples -- the two norm sums are all-reduced inside every corrector step."""
    rs = np.random.RandomState(13)
    B, N = 45, 6                                                     # 23 + 22 samples
    z0 = torch.tensor((rs.standard_normal((B, 63)) * 0.3).astype(np.float32))
    noise = torch.tensor(rs.standard_normal((N, 2, B, 63)).astype(np.float32))
    ref = _langevin_run(z0, noise, 1, 0).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_langevin, args=(r, 2, port, q, z0, noise)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.concatenate([res[0], res[1]], axis=0)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


def test_bench_runs_as_two_ranks_launched_by_torchrun():
    """The driver's multi-GPU invocation of bench.py (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N),
    with two fresh ranks sharing cuda:0 over gloo: rendezvous happens before any GPU call, rank 0 prints ONE JSON line with
    the contract's keys for a dp2 run, the per-GPU batch is half the global one, and the loss is finite."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DPOSER_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", DPOSER_BENCH_N1_LEG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--global-batch", "8192", "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 8192 and j["config"]["per_gpu_batch"] == 4096
    assert j["value"] > 0 and abs(j["value"] - 8192 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
    assert np.isfinite(j["extra"]["train_loss_last_step"]) and j["extra"]["nonfinite_gradient_steps_dropped"] == 0
    assert j["roofline"]["bound"] == "mfma" and "traffic_source" in j["roofline"]
    # the data-parallel step is attributable: what the collectives moved and what of them the step waited for
    dp = j["extra"]["dp"]
    assert dp["world"] == 2 and dp["backend"] == "gloo" and dp["steps"] == 3
    assert dp["collectives_per_step"] >= 2 and dp["gradient_buckets"] >= dp["collectives_per_step"]
    live = dp["flat_gradient_bytes"] - 4 * (1024 * 1024 + 1024)         # every gradient byte but the dead pre_dense_cond range travels
    assert dp["allreduce_bytes_per_step"] == live, (dp["allreduce_bytes_per_step"], live)
    assert 0.0 <= dp["exposed_allreduce_ms_per_step"] <= dp["rank_ms_per_step"]
    assert abs(dp["rank_compute_ms_per_step"] + dp["exposed_allreduce_ms_per_step"] - dp["rank_ms_per_step"]) < 1e-6
    # ... and self-validating: the communicator's own size, which device every rank ran on, every rank's step time, and the N = 1 step
    # of the same build on every rank's GPU (here both ranks share cuda:0 -- which the line must SAY)
    rk = j["extra"]["ranks"]
    assert rk["communicator"]["world_size_reported_by_the_process_group"] == 2 and rk["communicator"]["launcher_world_size"] == 2
    assert rk["communicator"]["backend"] == "gloo" and rk["communicator"]["rccl_version"] is None
    assert [r["rank"] for r in rk["per_rank"]] == [0, 1] and all(r["per_gpu_batch"] == 4096 for r in rk["per_rank"])
    assert all("MI3" in r["device_name"] or "Instinct" in r["device_name"] or r["device_name"] for r in rk["per_rank"])
    assert rk["distinct_devices"] == 1 and rk["all_ranks_on_distinct_devices"] is False
    assert all(r["own_ms_per_step"] > 0 and r["own_ms_per_step"] <= j["ms_per_step"] * 1.0001 for r in rk["per_rank"])
    n1 = rk["n1_same_build"]
    assert len(n1["ms_per_step_per_rank"]) == 2 and all(x > 0 for x in n1["ms_per_step_per_rank"])
    assert abs(n1["poses_per_s_rank0"] - 8192 / (n1["ms_per_step_per_rank"][0] * 1e-3)) / n1["poses_per_s_rank0"] < 1e-6


def test_bench_attributes_the_step_of_a_forced_one_rank_rccl_group():
    """bench.py through a ONE-rank process group over the real RCCL transport (DPOSER_DIST_FORCE_COLLECTIVES=1): `extra.dp` reports the
    bytes handed to ncclAllReduce, the number of collectives and the exposed wait (an event pair on the compute stream) -- what makes
    the first 8-GPU run attributable."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("DPOSER_DIST_BACKEND", "DPOSER_ZERO1")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DPOSER_DIST_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--global-batch", "8192",
                        "--no-extra", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    dp = j["extra"]["dp"]
    assert dp["backend"] == "nccl" and dp["world"] == 1 and dp["steps"] == 4
    assert dp["allreduce_bytes_per_step"] == dp["flat_gradient_bytes"] - 4 * (1024 * 1024 + 1024)
    assert 0.0 <= dp["exposed_allreduce_ms_per_step"] < dp["rank_ms_per_step"]
    rk = j["extra"]["ranks"]
    assert rk["communicator"]["backend"] == "nccl" and rk["communicator"]["rccl_version"] and rk["communicator"]["forced_one_rank_collectives"] is True
    assert rk["communicator"]["world_size_reported_by_the_process_group"] == 1 and rk["all_ranks_on_distinct_devices"] is True


def test_nonfinite_gradient_step_is_dropped_on_the_device():
    """A NaN in the batch poisons the gradient: the fused update must leave parameters, Adam moments and EMA untouched and count
    the dropped step, without any host synchronisation in the step itself (NaN-loss guard)."""
    from gpu_common import make_model
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    cfg, m, p = make_model(3, precision="bf16", dropout=0.1)
    cfg.optim.warmup = 0
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    opt = losses.get_optimizer(cfg, m.parameters())
    ema = ExponentialMovingAverage(m.parameters(), 0.9999)
    state = dict(model=m, optimizer=opt, ema=ema, step=0)
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    x = torch.randn(256, 63, device="cuda:0")
    step_fn(state, x)
    assert opt.nonfinite_steps() == 0
    before = (m.flat_params().clone(), opt._flat_m.clone(), opt._flat_v.clone(), ema._flat_shadow.clone())
    bad = x.clone()
    bad[17, 5] = float("nan")
    out = step_fn(state, bad)
    assert not np.isfinite(float(out["step_loss"]))
    assert opt.nonfinite_steps() == 1
    after = (m.flat_params(), opt._flat_m, opt._flat_v, ema._flat_shadow)
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    step_fn(state, x)                                      # and training continues
    assert opt.nonfinite_steps() == 1 and not torch.equal(m.flat_params(), before[0])


def _worker_zero1(rank, world, port, q, model_seed, batch, t, z, n_steps):
    os.environ["DPOSER_ZERO1"] = "1"
    _worker(rank, world, port, q, model_seed, batch, t, z, n_steps)


def test_zero1_sharded_step_equals_replicated_step():
    """DPOSER_ZERO1=1: reduce-scatter of the gradient, clip from one all-reduced squared norm, Adam / EMA on the owned range only,
    all-gather of the parameters -- must track the replicated (all-reduce) step and the single-process step on the whole batch."""
    rs = np.random.RandomState(5)
    n_steps, B = 3, 256
    batch = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    t = torch.tensor(rs.uniform(1e-3, 1.0, (n_steps, B)).astype(np.float32))
    z = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    ref, ref_mom, ref_losses = _steps(9, batch, t, z, n_steps, 1, 0, "fp32")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_zero1, args=(r, 2, port, q, 9, batch, t, z, n_steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    f0, f1 = res[0][1], res[1][1]
    assert np.array_equal(f0, f1)                                   # every rank holds the same gathered parameters
    assert np.abs(f0 - ref.numpy()).max() < 2e-5
    # moments live only on the owner of a range: the two halves together are the replicated first moment
    n = f0.size
    per = (-(-n // 2) + 3) // 4 * 4
    mom = np.concatenate([res[0][2][:per], res[1][2][per:]])
    assert np.linalg.norm(mom - ref_mom.numpy()) / np.linalg.norm(ref_mom.numpy()) < 1e-4
    assert np.abs(res[0][2][per:]).max() == 0.0 and np.abs(res[1][2][:per]).max() == 0.0
    mean_losses = [(a + b) / 2 for a, b in zip(res[0][3], res[1][3])]
    assert np.allclose(mean_losses, ref_losses, rtol=2e-5)


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it (the shape of the driver's single-GPU command with another N):
    bench.py must start the two ranks itself, as children of a parent that has not touched the GPU, and rank 0's JSON line must
    say n_gpus = 2.  (One GPU here: DPOSER_BENCH_ALLOW_SHARED_GPU=1 stacks the ranks on cuda:0 over gloo.)  Without that
    override and with fewer devices than ranks it must refuse with a non-zero exit code instead of reporting one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--global-batch", "4096",
           "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=dict(base, DPOSER_DIST_BACKEND="gloo", DPOSER_BENCH_ALLOW_SHARED_GPU="1"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["per_gpu_batch"] == 2048
    assert np.isfinite(j["extra"]["train_loss_last_step"])
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, cwd=root, env=base, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


_RCCL_SCRIPT = r'''
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["DPOSER_ROOT"], "tests", "golden"))
sys.path.insert(0, os.path.join(os.environ["DPOSER_ROOT"], "tests"))
sys.path.insert(0, os.environ["DPOSER_ROOT"])
from gpu_common import make_model
from dposer_amd import distributed as ddp
from dposer_amd.algorithms.advanced import losses, sde_lib
from dposer_amd.algorithms.ema import ExponentialMovingAverage

def run(n_steps, zero1):
    cfg, m, p = make_model(9, precision="fp32", dropout=0.0)
    cfg.optim.warmup = 0
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    opt = losses.get_optimizer(cfg, m.parameters())
    opt.zero1 = zero1
    ema = ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate)
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    state = dict(model=m, optimizer=opt, ema=ema, step=0)
    rs = np.random.RandomState(5)
    for i in range(n_steps):
        x = torch.tensor(rs.standard_normal((192, 63)).astype(np.float32)).cuda()
        t = torch.tensor(rs.uniform(1e-3, 1.0, 192).astype(np.float32)).cuda()
        z = torch.tensor(rs.standard_normal((192, 63)).astype(np.float32)).cuda()
        step_fn(state, x, t=t, z=z)
    torch.cuda.synchronize()
    opt.gather_state()                                      # (no-op unless the ZeRO-1 step sharded the moments)
    sd = opt.state_dict()
    return m.flat_params().detach().cpu().numpy().copy(), opt._flat_m.detach().cpu().numpy().copy(), sd

assert not ddp.dp_active()
ref, ref_m, _ = run(3, False)                               # no process group: the plain single-process step
rk, ws, lr = ddp.init_from_env()                            # WORLD_SIZE=1 + DPOSER_DIST_FORCE_COLLECTIVES=1 -> a one-rank RCCL group
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl" and ddp.dp_active()
a, a_m, _ = run(3, False)                                   # bucket events -> communication stream -> ncclAllReduce per bucket
b, b_m, sd = run(3, True)                                   # reduce_scatter_tensor / all_gather_into_tensor (padded: 8277567 % 4 != 0 handled for any world)
cnt = torch.tensor([5.0], dtype=torch.float64)              # a CPU tensor handed to the helper under RCCL (the Langevin global-batch count)
ddp.all_reduce_sum_(cnt)
flat = torch.arange(10, dtype=torch.float32, device="cuda")
bounds = ddp.zero1_bounds(10, 1)
ddp.reduce_scatter_flat_(flat, bounds); ddp.all_gather_flat_(flat, bounds)
print(json.dumps({"bucketed_max": float(np.abs(a - ref).max()), "bucketed_m": float(np.abs(a_m - ref_m).max()),
                  "zero1_max": float(np.abs(b - ref).max()), "zero1_m": float(np.abs(b_m - ref_m).max()), "cnt": float(cnt[0]),
                  "flat_ok": bool(torch.equal(flat.cpu(), torch.arange(10, dtype=torch.float32))),
                  "sd_steps": sorted({int(v["step"]) for v in sd["state"].values()})}))
dist.destroy_process_group()
'''


def test_single_rank_group_drives_the_real_rccl_transport():
    """Every other distributed test runs over gloo (two ranks cannot share one GPU under RCCL).  This one forms a ONE-rank
    process group with backend "nccl" (= RCCL) and DPOSER_DIST_FORCE_COLLECTIVES=1, which disables the world-size-1 shortcuts:
    the training step then runs its bucketed all-reduce (HIP event per bucket -> communication stream -> ncclAllReduce), the
    ZeRO-1 variant its reduce-scatter / all-gather, and the helper stages a CPU tensor through the device -- all over the real
    transport.  With one rank every collective is the identity, so the results must equal the plain single-process run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("DPOSER_DIST_BACKEND", "DPOSER_ZERO1")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DPOSER_DIST_FORCE_COLLECTIVES="1", DPOSER_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["bucketed_max"] == 0.0 and j["bucketed_m"] == 0.0            # identity collectives: bit-identical
    assert j["zero1_max"] < 2e-6 and j["zero1_m"] < 1e-6                  # (separate squared-norm launch: same value, other summation order)
    assert j["cnt"] == 5.0 and j["flat_ok"] and j["sd_steps"] == [3]


def _worker_zero1_ckpt(rank, world, port, q, model_seed, batch, t, z):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DPOSER_DIST_BACKEND="gloo", DPOSER_ZERO1="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_common import make_model
    from dposer_amd import distributed as ddp
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    import torch.distributed as dist
    ddp.init_from_env()
    torch.cuda.set_device(0)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)

    def fresh():
        cfg, m, p = make_model(model_seed, precision="fp32", dropout=0.0)
        cfg.optim.warmup = 0
        opt = losses.get_optimizer(cfg, m.parameters())
        ema = ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate)
        return cfg, dict(model=m, optimizer=opt, ema=ema, step=0)

    cfg, state = fresh()
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    lo, hi = ddp.shard_bounds(batch.shape[1], world, rank)

    def go(st, i):
        step_fn(st, batch[i, lo:hi].cuda(), t=t[i, lo:hi].cuda(), z=z[i, lo:hi].cuda())

    for i in range(2):
        go(state, i)
    import copy
    # the moments are sharded: state_dict() refuses (an implicit collective would hang a 'rank 0 saves' checkpoint) until EVERY rank
    # has gathered them; deepcopy = what torch.save would have written at this point (state dicts hold references to the live
    # buffers, which the next step overwrites)
    try:
        state["optimizer"].state_dict()
        raise AssertionError("state_dict() of sharded moments must raise")
    except RuntimeError as e:
        assert "gather_state" in str(e)
    state["optimizer"].gather_state()
    ck = copy.deepcopy({"model": state["model"].state_dict(), "opt": state["optimizer"].state_dict(), "ema": state["ema"].state_dict(),
                        "step": state["step"]})
    mom_full = state["optimizer"]._flat_m.detach().cpu().numpy().copy()
    go(state, 2)                                                                       # continue the original run
    cfg2, st2 = fresh()                                                                # ... and a restored one
    st2["model"].load_state_dict(ck["model"])
    st2["optimizer"].load_state_dict(ck["opt"])
    st2["ema"].load_state_dict(ck["ema"])
    st2["step"] = ck["step"]
    go(st2, 2)
    torch.cuda.synchronize()
    q.put((rank, state["model"].flat_params().detach().cpu().numpy(), st2["model"].flat_params().detach().cpu().numpy(), mom_full))
    ddp.barrier()
    dist.destroy_process_group()


def test_zero1_checkpoint_gathers_the_moments_and_resumes():
    """A ZeRO-1 run keeps each Adam moment on the rank that owns its range.  ``optimizer.gather_state()`` (an explicit collective,
    every rank) makes ``state_dict()`` the state of the replicated run -- ``state_dict()`` on ungathered state raises instead of
    entering a collective implicitly -- so that a checkpoint taken mid-run restores into a run that continues exactly like the
    original one."""
    rs = np.random.RandomState(5)
    B = 256
    batch = torch.tensor(rs.standard_normal((3, B, 63)).astype(np.float32))
    t = torch.tensor(rs.uniform(1e-3, 1.0, (3, B)).astype(np.float32))
    z = torch.tensor(rs.standard_normal((3, B, 63)).astype(np.float32))
    ref, ref_mom, _ = _steps(9, batch[:2], t[:2], z[:2], 2, 1, 0, "fp32")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_zero1_ckpt, args=(r, 2, port, q, 9, batch, t, z)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, cont, resumed, mom in res:
        assert np.array_equal(cont, resumed)                                           # restored run == uninterrupted run, bit for bit
        assert np.linalg.norm(mom - ref_mom.numpy()) / np.linalg.norm(ref_mom.numpy()) < 1e-4   # gathered moments = replicated moments
    assert np.array_equal(res[0][1], res[1][1])


def test_two_rank_autograd_step_averages_the_gradient():
    """The steps that are NOT the fused pipeline (likelihood weighting here; the auxiliary-loss step shares the code) run
    loss.backward() through autograd.  Under data parallelism their ``p.grad`` must be all-reduced (mean) before clip + Adam --
    the reference runs them under nn.DataParallel, whose gradient covers the global batch -- so two ranks on half batches track the
    single-process step on the whole batch and stay bit-identical to each other (they used to update unsynchronised replicas)."""
    rs = np.random.RandomState(6)
    n_steps, B = 2, 128
    batch = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    t = torch.tensor(rs.uniform(0.05, 1.0, (n_steps, B)).astype(np.float32))
    z = torch.tensor(rs.standard_normal((n_steps, B, 63)).astype(np.float32))
    ref, ref_mom, ref_losses = _steps(9, batch, t, z, n_steps, 1, 0, "fp32", True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 9, batch, t, z, n_steps, "fp32", True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    f0, f1 = res[0][1], res[1][1]
    assert np.array_equal(f0, f1)                                            # replicas stay in step
    rel = np.linalg.norm(f0 - ref.numpy()) / np.linalg.norm(ref.numpy())
    assert rel < 2e-5, rel                                                    # == the whole-batch step (measured 1e-7)
    assert abs(0.5 * (res[0][3][0] + res[1][3][0]) - ref_losses[0]) < 2e-5 * abs(ref_losses[0])
