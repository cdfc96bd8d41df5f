"""Deterministic test-signal closed forms (test infrastructure).
Restates the parts of src/signals/Signals.jl:14-140, 188-235 that need no RNG
(:sin, :sin_decay, :abs_cos_power_p8).  The seeded kinds (:random, :multi_sin,
:multi_sin_exp) draw from Julia's Xoshiro stream, which cannot be reproduced
outside Julia; for those the oracle uses numpy's PCG64 with the same structure.
"""
from __future__ import annotations

import numpy as np


def _normalize(v):
    return v / np.linalg.norm(v)


def generate_signal(n, kind="sin", dt=None, freq=None, **kw):
    N = 2 ** n
    j = np.arange(N, dtype=np.float64)
    if kind == "random":                                       # Signals.jl:93-96, 197-200
        return np.random.default_rng(kw.get("seed", 1234)).standard_normal(N)
    fv = 2 * np.pi if freq is None else freq                   # :203-204
    vec = not np.isscalar(fv)
    fv = np.asarray(fv, dtype=np.float64) if vec else float(fv)
    if dt is None:                                             # :207-216
        fmax = float(np.max(np.abs(fv)))
        dt = 1.0 if fmax == 0 else 1.0 / (fmax * N)
    if kind == "sin":                                          # :14-25, 47-62
        noise = kw.get("noise_level", 0.0)
        if vec:
            phase = np.asarray(kw.get("phase", np.zeros(len(fv))), dtype=np.float64)
            if len(phase) != len(fv):
                raise ValueError("Frequency and phase vectors must be of the same length.")
            x = sum(np.sin(w * dt * j + p) for w, p in zip(fv, phase))
        else:
            x = np.sin(fv * dt * j + kw.get("phase", 0.0))
        if noise:
            x = x + noise * np.random.default_rng(kw.get("seed", 0)).standard_normal(N)
        return x
    if kind == "sin_decay":                                    # :99-140
        dr = kw["decay_rate"]
        if vec:
            dr = np.asarray(dr, dtype=np.float64)
            if len(dr) != len(fv):
                raise ValueError("Frequency and decay_rate vectors must be of the same length.")
            phase = kw.get("phase")
            phase = np.zeros(len(fv)) if phase is None else np.asarray(phase, dtype=np.float64)
            return sum(np.sin(w * dt * j + p) * np.exp(-l * dt * j) for w, l, p in zip(fv, dr, phase))
        return np.sin(fv * dt * j + kw.get("phase", 0.0)) * np.exp(-float(dr) * dt * j)
    if kind == "abs_cos_power_p8":                             # :87-90
        return np.abs(np.cos(2 * np.pi * dt * j)) ** kw.get("power", 0.8)
    if kind in ("multi_sin", "multi_sin_exp"):                 # :27-45, 64-85 (own RNG stream)
        nt = kw.get("n_terms", 10)
        ak = _normalize(np.random.default_rng(kw.get("seed_amp", 1001)).random(nt))
        wk = kw.get("omega_scale", 40.0) * dt * (np.random.default_rng(kw.get("seed_freq", 2002)).random(nt) - 0.5)
        if kind == "multi_sin":
            return sum(ak[k] * np.sin(wk[k] * j) for k in range(nt))
        lk = -kw.get("lambda_scale", 2.0) * dt * np.random.default_rng(kw.get("seed_decay", 4004)).random(nt)
        return sum(ak[k] * np.sin(wk[k] * j) * np.exp(lk[k] * j) for k in range(nt))
    raise ValueError(f"Unsupported signal kind: {kind}")
