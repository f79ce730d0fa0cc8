"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference has no DDP training (run/train.py is single device) and shards evaluation with a
gloo process group (run/completion.py:83-92, lib/dataset/EvaSampler.py).  On an 8 x MI355X node the
training step is data parallel over the batch: every rank computes the DSM gradient of its
contiguous B/G shard into ONE flat fp32 buffer (8.28 M floats = 33.1 MB).  The backward pass
finishes that buffer in five contiguous buckets (last GN layer + post_dense first, 6.3 MB each,
dposer_scorefc_grad_buckets) and records a HIP event per bucket; ``all_reduce_buckets_`` sums each
bucket on a side stream as soon as its event fires, so the collective of layer l overlaps the
dgrad/wgrad GEMMs of layers l-1..0 (at B/G = 8192 the step is ~1.3 ms and an un-overlapped 33 MB
ring all-reduce ~0.3 ms; xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so 6 MB messages
still use every link).  The 1/world factor and the global-norm clip are folded into the fused
Adam/EMA kernel (grad_scale).  Sampling, completion and FK shard the batch with no collective.
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def _forced():
    """DPOSER_DIST_FORCE_COLLECTIVES=1: run every collective even in a one-rank group.  A single-GPU box can then drive the
    real RCCL transport (backend "nccl", world size 1) through exactly the code a multi-GPU job runs -- bucket events, the
    communication stream, reduce-scatter / all-gather -- instead of the world-size-1 shortcuts (tests/test_gpu_distributed.py)."""
    return os.environ.get("DPOSER_DIST_FORCE_COLLECTIVES") == "1"


_LOCAL_ONLY = 0


def dp_active():
    """True when collectives have to run: a process group with more than one rank (or a forced one-rank group) -- and no
    ``local_only()`` block is open."""
    return _LOCAL_ONLY == 0 and is_initialized() and (dist.get_world_size() > 1 or _forced())


class local_only:
    """``with local_only():`` -- inside a data-parallel job, run the enclosed steps as a single-process job would (no collective, no
    per-rank Philox offset): bench.py times the N = 1 step of the SAME build on every rank's GPU this way, next to the N-rank number.
    Every rank must leave the block before the next collective; parameters that were stepped inside it have diverged between the ranks
    (the caller restores or re-broadcasts them)."""

    def __enter__(self):
        global _LOCAL_ONLY
        _LOCAL_ONLY += 1
        return self

    def __exit__(self, *exc):
        global _LOCAL_ONLY
        _LOCAL_ONLY -= 1
        return False


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def rank():
    return dist.get_rank() if is_initialized() else 0


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_*).  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or _forced()) and not is_initialized():
        if backend is None:
            backend = os.environ.get("DPOSER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(lr % max(torch.cuda.device_count(), 1))
            # fail fast: a failed / timed-out RCCL collective aborts the process instead of hanging its peers
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        import datetime
        timeout = datetime.timedelta(seconds=int(os.environ.get("DPOSER_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=rk, world_size=ws, timeout=timeout)
    return rk, ws, lr


def run_fail_fast(fn, *args, **kwargs):
    """Entry-point wrapper for multi-process jobs: any exception on this rank (a collective error, a device fault, a failed
    check) prints its traceback and ends the PROCESS with a non-zero code at once -- no destroy_process_group() that would
    wait for peers stuck in a collective -- so the launcher (torchrun) tears the job down.  The reference's train loop swallows
    exceptions (train.py:243,406-407); a data-parallel job must not."""
    import sys
    import traceback
    try:
        return fn(*args, **kwargs)
    except SystemExit:
        raise
    except BaseException:
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


def all_reduce_sum_(flat: torch.Tensor) -> int:
    """In-place SUM all-reduce; returns the world size (the caller divides).  The tensor travels on whatever the backend
    can move: RCCL ("nccl") only reduces device tensors and gloo test rigs reduce on the host, so a tensor on the other side
    is staged through a copy (a CPU tensor handed to an NCCL collective raises "No backend type associated with device type
    cpu" -- that was a crash of every data-parallel Langevin run)."""
    if not dp_active():
        return 1
    backend = dist.get_backend()
    if backend == "nccl" and not flat.is_cuda:
        dev = flat.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(dev, op=dist.ReduceOp.SUM)
        flat.copy_(dev)
    elif backend == "gloo" and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return dist.get_world_size()


_COMM_STREAMS = {}


def all_reduce_buckets_(flat: torch.Tensor, buckets, wait_bucket=None) -> int:
    """In-place SUM all-reduce of ``flat`` bucket by bucket (``buckets`` = [(lo, hi)] in the order the producer
    finishes them).  On a GPU the collectives are issued from a dedicated communication stream that first
    waits for the producer's per-bucket event (``wait_bucket(i, raw_stream_handle)``), so bucket i is reduced
    while later buckets are still being computed; the caller's stream is made to wait for all of them before
    returning.  Returns the world size (the caller divides)."""
    if not dp_active():
        return 1
    works = []
    if flat.is_cuda:
        key = flat.device.index
        comm = _COMM_STREAMS.get(key)
        if comm is None:
            comm = _COMM_STREAMS[key] = torch.cuda.Stream(device=flat.device)
        with torch.cuda.stream(comm):
            for i, (lo, hi) in enumerate(buckets):
                if wait_bucket is not None:
                    wait_bucket(i, comm.cuda_stream)
                else:
                    comm.wait_stream(torch.cuda.default_stream(flat.device))
                works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
    else:
        for lo, hi in buckets:
            works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()                      # NCCL: the current stream waits for the collective; gloo: host wait
    return dist.get_world_size()


# ---- attribution of a data-parallel step (bench.py `extra.dp`): what the collectives moved and what of them the step had to wait for ----
_STATS = None


def stats_enable(on=True):
    """Start (or stop) recording, per data-parallel step: bytes handed to the all-reduce, number of collectives and the time the
    compute stream stalled in ``StreamedAllReduce.finish()`` (an event pair around the waits: the EXPOSED part of the communication --
    everything else ran under the backward pass)."""
    global _STATS
    _STATS = {"steps": 0, "bytes": 0, "collectives": 0, "events": [], "host_wait_s": 0.0} if on else None


def stats_collect():
    """{steps, allreduce_bytes_per_step, collectives_per_step, exposed_allreduce_ms_per_step} of the steps since stats_enable(); None
    when nothing was recorded.  Synchronises the device (reads the event pairs)."""
    st = _STATS
    if not st or st["steps"] == 0:
        return None
    exposed = st["host_wait_s"] * 1e3
    if st["events"]:
        torch.cuda.synchronize()
        exposed += sum(a.elapsed_time(b) for a, b in st["events"])
    n = st["steps"]
    return {"steps": n, "allreduce_bytes_per_step": st["bytes"] / n, "collectives_per_step": st["collectives"] / n,
            "exposed_allreduce_ms_per_step": exposed / n,
            "how": "event pair on the compute stream around the waits for the bucket collectives (gloo: host wall time of the waits)"}


class StreamedAllReduce:
    """All-reduce (SUM) of ``flat`` range by range WHILE the producer is still queueing work: the fused backward pass announces
    every group of final gradient buckets from inside its C call (dposer_dsm_loss_fwd_bwd_notify: one recorded event + the merged
    flat ranges); ``on_final`` makes the communication stream wait for that event and enqueues the collectives of those ranges at
    once.  The host-side cost of a torch.distributed call (tens of microseconds) is then paid while the GPU still has the rest of
    the backward pass queued, not after it -- with one rank per GPU at 8192 poses the step is ~0.6 ms, six collectives issued after
    the C call returned were ~0.15 ms of exposed host time -- and every collective overlaps with the layers still being
    differentiated.  ``finish()`` makes the current stream wait for all of them and returns the world size (the caller divides)."""

    def __init__(self, flat: torch.Tensor, wait_event):
        self.flat, self.wait_event = flat, wait_event
        self.works, self.error, self.ranges = [], None, []
        self.comm = None
        if flat.is_cuda:
            key = flat.device.index
            self.comm = _COMM_STREAMS.get(key)
            if self.comm is None:
                self.comm = _COMM_STREAMS[key] = torch.cuda.Stream(device=flat.device)

    def on_final(self, ranges, event):
        """``ranges``: [(lo, hi)] final once ``event`` (raw handle, recorded on the producer's stream) has fired."""
        try:                                  # (called through ctypes from inside the C call: an exception must not unwind through it)
            self.ranges.extend(ranges)
            if self.comm is not None:
                self.wait_event(self.comm.cuda_stream, event)
                with torch.cuda.stream(self.comm):
                    for lo, hi in ranges:
                        self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            else:
                for lo, hi in ranges:
                    self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        except BaseException as e:            # noqa: BLE001 -- re-raised by finish()
            self.error = e

    def finish(self, expect=None) -> int:
        """``expect``: the bucket ranges the producer must have announced (every one covered by an announced range), checked after
        the wait -- a bucket that was never announced was never reduced."""
        st = _STATS
        ev = None
        if st is not None:
            if self.comm is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            else:
                import time
                t0 = time.perf_counter()
        for w in self.works:                  # first: whatever was enqueued is joined, also on the error path -- the caller must
            w.wait()                          # not see ``flat`` while a collective still writes it (NCCL: the current stream waits; gloo: host wait)
        if st is not None:
            if ev is not None:
                ev[1].record()
                st["events"].append(ev)
            else:
                st["host_wait_s"] += time.perf_counter() - t0
            st["steps"] += 1
            st["collectives"] += len(self.works)
            st["bytes"] += sum(hi - lo for lo, hi in self.ranges) * self.flat.element_size()
        if self.error is not None:
            raise self.error
        if expect is not None:
            for lo, hi in expect:
                if hi > lo and not any(a <= lo and hi <= b for a, b in self.ranges):
                    raise RuntimeError(f"gradient bucket [{lo}, {hi}) was never announced as final: not all-reduced")
        return dist.get_world_size()


class StreamedReduceToOwners(StreamedAllReduce):
    """The ZeRO-1 form of StreamedAllReduce: ``bounds[r]`` = the contiguous range of ``flat`` whose optimizer state rank r owns.  Every
    announced range is cut at the ownership boundaries and each piece is SUM-reduced to its owner (``dist.reduce``) from the
    communication stream, while the backward pass is still running -- the reduce-scatter of the sharded step, issued bucket by bucket
    under the backward instead of as one collective after it (half the wire bytes of the all-reduce the replicated step issues).
    After ``finish()`` rank r holds the reduced gradient of ``bounds[r]``; other ranges are unspecified."""

    def __init__(self, flat, wait_event, bounds):
        super().__init__(flat, wait_event)
        self.bounds = list(bounds)

    def _pieces(self, ranges):
        for lo, hi in ranges:
            for r, (a, b) in enumerate(self.bounds):
                p, q = max(lo, a), min(hi, b)
                if q > p:
                    yield r, p, q

    def on_final(self, ranges, event):
        try:
            self.ranges.extend(ranges)
            if self.comm is not None:
                self.wait_event(self.comm.cuda_stream, event)
                with torch.cuda.stream(self.comm):
                    for r, p, q in self._pieces(ranges):
                        self.works.append(dist.reduce(self.flat[p:q], dst=r, op=dist.ReduceOp.SUM, async_op=True))
            else:
                for r, p, q in self._pieces(ranges):
                    self.works.append(dist.reduce(self.flat[p:q], dst=r, op=dist.ReduceOp.SUM, async_op=True))
        except BaseException as e:            # noqa: BLE001 -- re-raised by finish()
            self.error = e


def zero1_bounds(n: int, num_replicas: int):
    """Contiguous ownership ranges of a flat buffer of n elements for the sharded optimiser step: equal sizes rounded up to a
    multiple of 4 floats (16-byte aligned shard starts), the last rank takes what is left (it may be shorter: the collectives
    below run on a staging buffer padded to num_replicas equal shards, so the single-collective path is always taken)."""
    per = -(-n // num_replicas)
    per = (per + 3) // 4 * 4
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(num_replicas)]


def _equal_shards(flat, bounds):
    """(per, padded) when ``bounds`` are the zero1_bounds of ``flat``: shards of ``per`` elements, the last one possibly short."""
    per = bounds[0][1] - bounds[0][0]
    ok = per > 0 and all(lo == r * per or lo == flat.numel() for r, (lo, hi) in enumerate(bounds)) and bounds[-1][1] == flat.numel()
    return per if ok else 0, per * len(bounds)


def reduce_scatter_flat_(flat: torch.Tensor, bounds):
    """SUM-reduce ``flat`` so that rank r ends with the reduced values of its range ``bounds[r]`` (other ranges: unspecified).
    RCCL: ONE reduce_scatter over equal chunks -- straight from ``flat`` when the ranges tile it exactly, else from a copy
    padded with zeros to world x per elements (one extra 33 MB device copy instead of `world` reduce calls); gloo has no
    reduce-scatter: one reduce per range."""
    if not dp_active():
        return 1
    ws, rk = dist.get_world_size(), dist.get_rank()
    per, padded = _equal_shards(flat, bounds)
    if dist.get_backend() == "nccl" and per:
        lo, hi = bounds[rk]
        src = flat
        if padded != flat.numel():
            src = torch.zeros(padded, dtype=flat.dtype, device=flat.device)
            src[:flat.numel()].copy_(flat)
        out = torch.empty(per, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM)
        flat[lo:hi].copy_(out[:hi - lo])
    else:
        works = [dist.reduce(flat[lo:hi], dst=r, op=dist.ReduceOp.SUM, async_op=True) for r, (lo, hi) in enumerate(bounds) if hi > lo]
        for w in works:
            w.wait()
    return ws


def all_gather_flat_(flat: torch.Tensor, bounds):
    """Every rank publishes its range ``bounds[rank]`` of ``flat``; afterwards all ranks hold all ranges."""
    if not dp_active():
        return
    _bump_param_epoch()
    rk = dist.get_rank()
    per, padded = _equal_shards(flat, bounds)
    if dist.get_backend() == "nccl" and per:
        lo, hi = bounds[rk]
        mine = torch.zeros(per, dtype=flat.dtype, device=flat.device)
        mine[:hi - lo].copy_(flat[lo:hi])
        if padded == flat.numel():
            dist.all_gather_into_tensor(flat, mine)
        else:
            full = torch.empty(padded, dtype=flat.dtype, device=flat.device)
            dist.all_gather_into_tensor(full, mine)
            flat.copy_(full[:flat.numel()])
    else:
        works = [dist.broadcast(flat[lo:hi], src=r, async_op=True) for r, (lo, hi) in enumerate(bounds) if hi > lo]
        for w in works:
            w.wait()


def broadcast_(flat: torch.Tensor, src=0):
    """Make every rank start from rank ``src``'s parameters."""
    if dp_active():
        dist.broadcast(flat, src=src)
        _bump_param_epoch()
    return flat


def _bump_param_epoch():
    """A collective rewrote a flat parameter buffer: c10d bumps no version counter, so the packed-weight freshness key
    (engine.param_state_key) would not notice -- bump the package's parameter epoch instead."""
    from . import _C
    _C.bump_param_epoch()


def shard_bounds(total: int, num_replicas: int, rank_: int):
    """Contiguous split used by the reference's DistributedEvalSampler (lib/dataset/EvaSampler.py:78-84):
    base = total // G, the first total % G ranks get one extra element."""
    base, extra = divmod(total, num_replicas)
    start = base * rank_ + min(rank_, extra)
    return start, start + base + (1 if rank_ < extra else 0)


def gather_metrics(all_results, dst=0):
    """End-of-evaluation reduction of run/completion.py:300-323: every rank holds ``all_results`` = a list (one entry per batch)
    of ``{metric name: per-sample values}``; rank ``dst`` receives them all, concatenates per metric and returns
    ``({name: mean over every sample of every rank}, {name: all values})``; other ranks get ``(None, None)``.
    Single process: the same arithmetic without a collective."""
    import numpy as np
    if is_initialized() and dist.get_world_size() > 1:
        collection = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(all_results, collection, dst=dst)
        if dist.get_rank() != dst:
            return None, None
        batches = [b for rank_results in collection for b in rank_results]
    else:
        batches = list(all_results)
    merged = {}
    for batch in batches:
        for key, value in batch.items():
            merged.setdefault(key, []).extend(np.asarray(value.detach().cpu() if torch.is_tensor(value) else value).reshape(-1).tolist())
    return {k: float(np.mean(np.array(v))) for k, v in merged.items()}, merged


def reduce_metric_means(all_results, device=None, names=None):
    """The means of run/completion.py:318-321 without shipping per-sample values: every rank folds its batches into one
    [n_metrics, 2] device tensor of (sum, count) in float64, ONE all-reduce (SUM) combines the ranks, and every rank gets
    ``{name: mean over every sample of every rank}``.  ``all_results``: list (one entry per batch) of {name: per-sample tensor}.
    ``names``: the fixed metric-name list of the evaluation.  Pass it whenever ranks can end up with NO batch (``drop_last``
    per rank on shards that differ by one element, fewer sequences than ranks): the all-reduced tensor must have the same shape
    on every rank, and a rank without results cannot learn the names from them (mismatched sizes hang or corrupt RCCL)."""
    if names is None:
        names = sorted({k for batch in all_results for k in batch})
    names = list(names)
    if device is None:
        device = next((v.device for batch in all_results for v in batch.values() if torch.is_tensor(v)), torch.device("cpu"))
    acc = torch.zeros(len(names), 2, dtype=torch.float64, device=device)
    for batch in all_results:
        for i, k in enumerate(names):
            if k in batch:
                v = torch.as_tensor(batch[k], device=device).reshape(-1).double()
                acc[i, 0] += v.sum()
                acc[i, 1] += v.numel()
    if dp_active():
        all_reduce_sum_(acc)
    acc = acc.cpu()
    return {k: float(acc[i, 0] / acc[i, 1]) if acc[i, 1] > 0 else float("nan") for i, k in enumerate(names)}


def barrier():
    if is_initialized():
        dist.barrier()
