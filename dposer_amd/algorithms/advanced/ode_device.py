"""Device-resident adaptive Runge-Kutta driver for the probability-flow ODE.

The reference integrates the probability-flow ODE with ``scipy.integrate.solve_ivp(method='RK45')`` (sampling.py:530,
likelihood.py:99): every right-hand-side evaluation moves the state host -> device -> host, and the step-size controller is
part of the result (``nfe`` is returned to the caller).  This module keeps the state, the seven stage derivatives and every
stage combination ON THE DEVICE (float64, like scipy's host arithmetic) and runs the SAME controller -- Dormand-Prince 5(4)
tableau, error norm ``||err / (atol + rtol max(|y|, |y_new|))||_rms``, step factor ``0.9 err^(-1/5)`` clamped to [0.2, 10],
Hairer's initial-step heuristic -- as published in scipy's ``integrate/_ivp/rk.py`` (``RK45``, ``rk_step``) and
``_ivp/common.py`` (``select_initial_step``, ``norm``), scipy 1.15.  One scalar (the error norm) crosses to the host per
attempted step -- instead of the whole state twice per right-hand-side evaluation, six times per step.

``solve_rk45(fun, t0, t1, y0, rtol, atol) -> (y(t1), nfev)``;  ``fun(t: float, y: float64 tensor [n]) -> float64 tensor [n]``.
``solve_fixed(fun, t0, t1, y0, n_steps, method='rk4')``: the fixed-step variant (no host synchronisation at all).
"""
import math

import torch

SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0

# Dormand-Prince 5(4) (Hairer, Norsett, Wanner, "Solving ODEs I", II.5; the coefficients scipy's RK45 tabulates)
_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0)
_A = ((),
      (1 / 5,),
      (3 / 40, 9 / 40),
      (44 / 45, -56 / 15, 32 / 9),
      (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
      (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656))
_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)
_E = (-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40)
ORDER_ERR = 4


def _combine(y, ks, coefs, scale):
    """[y +] scale * sum_j coefs[j] * ks[j], left to right in float64.  On the GPU: ONE launch of dposer_rk_combine_f64 (the torch
    expression is 2 len(ks) + 1 elementwise kernels per stage; at 8192 samples they cost as much as a fifth of a network
    evaluation); same bits either way (no fused multiply-add on either side)."""
    terms = [(k, c) for k, c in zip(ks, coefs) if c != 0.0]
    if ks[0].is_cuda and len(terms) <= 8:
        import ctypes as C
        from ... import _C
        out = torch.empty_like(ks[0])
        kp = (C.c_void_p * len(terms))(*[k.data_ptr() for k, _ in terms])
        cf = (C.c_double * len(terms))(*[float(c) for _, c in terms])
        assert all(k.is_contiguous() and k.dtype == torch.float64 for k, _ in terms) and (y is None or (y.is_contiguous() and y.dtype == torch.float64))
        _C.check(_C.lib().dposer_rk_combine_f64(_C.ptr(out), _C.ptr(y), kp, cf, len(terms), float(scale), out.numel(), _C.stream_ptr()),
                 "dposer_rk_combine_f64")
        return out
    acc = terms[0][0] * terms[0][1]
    for k, c in terms[1:]:
        acc = acc + k * c
    acc = acc * scale
    return acc if y is None else y + acc


def _rms(x):
    return float(torch.linalg.vector_norm(x) / math.sqrt(x.numel()))          # scipy _ivp/common.py norm(); the one host sync


def _initial_step(fun, t0, y0, t_bound, f0, direction, rtol, atol):
    """scipy _ivp/common.py select_initial_step (Hairer II.4); one extra right-hand-side evaluation."""
    interval = abs(t_bound - t0)
    if y0.numel() == 0:
        return math.inf
    if interval == 0.0:
        return 0.0
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval)
    f1 = fun(t0 + h0 * direction, y0 + h0 * direction * f0)
    d2 = _rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / (ORDER_ERR + 1))
    return min(100 * h0, h1, interval)


def solve_rk45(fun, t0, t1, y0, rtol=1e-3, atol=1e-6, max_steps=100000):
    """Integrate dy/dt = fun(t, y) from t0 to t1 (either direction); returns (y(t1), nfev) with scipy's evaluation count."""
    t0, t1 = float(t0), float(t1)
    y = y0.to(torch.float64).contiguous()
    nfev = [0]

    def f(t, yy):
        nfev[0] += 1
        return fun(t, yy).to(torch.float64).contiguous()

    direction = 1.0 if t1 >= t0 else -1.0
    fy = f(t0, y)
    h_abs = _initial_step(f, t0, y, t1, fy, direction, rtol, atol)
    t = t0
    K = [None] * 7
    exponent = -1.0 / (ORDER_ERR + 1)
    for _ in range(max_steps):
        if direction * (t - t1) >= 0:
            break
        min_step = 10 * abs(math.nextafter(t, direction * math.inf) - t)
        h_abs = max(h_abs, min_step)
        rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("solve_rk45: required step size is less than spacing between numbers")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - t1) > 0:
                t_new = t1
            h = t_new - t
            h_abs = abs(h)
            # rk_step: K[s] = fun(t + c_s h, y + h sum_j a_sj K[j]);  y_new = y + h sum_j b_j K[j];  K[6] = fun(t + h, y_new)
            K[0] = fy
            for s in range(1, 6):
                K[s] = f(t + _C[s] * h, _combine(y, K[:s], _A[s], h))
            y_new = _combine(y, K[:6], _B, h)
            f_new = f(t + h, y_new)
            K[6] = f_new
            err = _combine(None, K, _E, 1.0)
            scale = atol + torch.maximum(y.abs(), y_new.abs()) * rtol
            error_norm = _rms(err * h / scale)
            if error_norm < 1:
                factor = MAX_FACTOR if error_norm == 0 else min(MAX_FACTOR, SAFETY * error_norm ** exponent)
                if rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(MIN_FACTOR, SAFETY * error_norm ** exponent)
            rejected = True
        t, y, fy = t_new, y_new, f_new
    else:
        raise RuntimeError("solve_rk45: step budget exhausted")
    return y, nfev[0]


def solve_fixed(fun, t0, t1, y0, n_steps, method="rk4"):
    """Fixed-step explicit integration (classical RK4 or Euler) with no host synchronisation: n_steps x {4, 1} evaluations."""
    t0, t1 = float(t0), float(t1)
    y = y0.to(torch.float64).contiguous()
    h = (t1 - t0) / n_steps
    nfev = 0
    for i in range(n_steps):
        t = t0 + i * h
        if method == "euler":
            y = y + h * fun(t, y).to(torch.float64)
            nfev += 1
        elif method == "rk4":
            k1 = fun(t, y).to(torch.float64).contiguous()
            k2 = fun(t + h / 2, _combine(y, [k1], [1.0], h / 2)).to(torch.float64).contiguous()
            k3 = fun(t + h / 2, _combine(y, [k2], [1.0], h / 2)).to(torch.float64).contiguous()
            k4 = fun(t + h, _combine(y, [k3], [1.0], h)).to(torch.float64).contiguous()
            y = _combine(y, [k1, k2, k3, k4], [1.0, 2.0, 2.0, 1.0], h / 6)
            nfev += 4
        else:
            raise ValueError(f"unknown fixed-step method {method!r}")
    return y, nfev
