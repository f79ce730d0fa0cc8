#!/usr/bin/env python3
"""Experiment (round 4): does a latency-bound training step gain from running as TWO independent half-batch chains on two streams?
At 8192 poses every GEMM of the step is one round of 128x128 tiles; the launches of one chain are strictly dependent, so ramp,
prologue and epilogue phases of every kernel leave the matrix pipe idle chip-wide.  Two half-batch chains have no dependence on
each other and can fill those phases.
    python tools/two_chain_ab.py [B]          # ms per forward+backward (no optimiser): one chain of B vs two chains of B/2
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

from dposer_amd import _C  # noqa: E402
from dposer_amd.algorithms.advanced import sde_lib  # noqa: E402
from dposer_amd.algorithms.advanced.sde_lib import sde_desc  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    iters = 200
    from gpu_common import make_model
    cfg, m, p = make_model(5, precision="bf16")
    m.train()
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    dev = torch.device("cuda", 0)
    eng = m._engine()
    flat = m.flat_params()
    packed = eng.packed(flat, with_backward=True, force=True)
    desc = sde_desc(sde)
    lib = eng.lib

    def ws_for(b):
        need = lib.dposer_scorefc_workspace_bytes(eng.h, b, _C.WS_TRAIN, 0)
        return torch.empty(need, dtype=torch.uint8, device=dev)

    x = torch.randn(B, 63, device=dev)
    freq, sig = eng.freq(dev), m.sigmas

    def call(xb, ws, fg, loss, stream, step):
        _C.check(lib.dposer_dsm_loss_fwd_bwd_bucketed(
            eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(xb), None, None, 1e-5, 7, step,
            _C.ptr(freq), _C.ptr(sig), _C.ptr(fg), _C.ptr(loss), xb.shape[0], None, 0, C.c_void_p(stream.cuda_stream)), "fwd_bwd")

    results = {}
    for label, nchain in (("one chain", 1), ("two chains", 2), ("one chain again", 1), ("two chains again", 2), ("four chains", 4)):
        b = B // nchain
        xs = [x[i * b:(i + 1) * b].contiguous() for i in range(nchain)]
        wss = [ws_for(b) for _ in range(nchain)]
        fgs = [torch.zeros(eng.num_params, device=dev) for _ in range(nchain)]
        ls = [torch.zeros(1, device=dev) for _ in range(nchain)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(nchain)]
        main_s = torch.cuda.current_stream()

        def one_iter(step):
            if nchain == 1:
                call(xs[0], wss[0], fgs[0], ls[0], main_s, step)
                return
            for s in streams:
                s.wait_stream(main_s)
            for i, s in enumerate(streams):
                call(xs[i], wss[i], fgs[i], ls[i], s, step)
            for s in streams:
                main_s.wait_stream(s)

        for it in range(10):
            one_iter(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(iters):
            one_iter(it)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        results[label] = ms
        print(f"B = {B}: {label}: {ms:.4f} ms per forward+backward (loss {float(sum(l[0] for l in ls)) / nchain:.4f})", flush=True)


if __name__ == "__main__":
    main()
