#!/usr/bin/env python3
"""Micro-benchmark of the projection kernel on the cfg5 shape (developer tool).
   variants: 0 = W-resident 16 waves x 16 rows, 2 = W-resident 8 waves x 32 rows, 1 = streaming-W.  modes: real = 5 distinct (M,64) terms; alias = 5 x the same term; lda0 = every row reads row 0 (compute only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tgcn_amd import _lib  # noqa: E402


def run(M, Kc, N, T, mode, variant, rounds=5):
    L = _lib.lib()
    dev = torch.device("cuda:0")
    if mode == "skew":      # 5 distinct terms carved from one buffer with different sub-page offsets (DRAM channel skew)
        big = torch.randn(T * (M * Kc + 65536 * 4), device=dev)
        terms = [big[t * (M * Kc + 65536 * 4) + t * 4160: t * (M * Kc + 65536 * 4) + t * 4160 + M * Kc].view(M, Kc) for t in range(T)]
    else:
        terms = [torch.randn(M, Kc, device=dev) for _ in range(T if mode == "real" else 1)]
    W = torch.randn(T * Kc, N, device=dev) / (T * Kc) ** 0.5
    out = torch.empty(M, N, device=dev)
    a = (C.c_void_p * T)(*[terms[i if mode in ("real", "skew") else 0].data_ptr() for i in range(T)])
    lda = (C.c_int64 * T)(*[0 if mode == "lda0" else Kc for _ in range(T)])
    _lib.check(L.tgcn_set_tuning(b"project_variant", variant))
    ts = []
    for r in range(rounds + 1):
        _lib.profile_start(16)
        if True:
            _lib.check(L.tgcn_cheb_project_f32(_lib.stream_ptr(), M, Kc, N, T, a, lda, _lib.ptr(W), None, 0, M, 1, 0, _lib.ptr(out), N))
        prof = _lib.profile_stop(16)
        if r:
            ts.append(sum(ms for k, ms in prof if k == 2))
    _lib.check(L.tgcn_set_tuning(b"project_variant", 0))
    t = float(np.median(ts))
    print("M=%d Kc=%d N=%d T=%d %-5s variant %d: %.3f ms  %.1f TFLOP/s  %.2f TB/s" % (
        M, Kc, N, T, mode, variant, t, 2.0 * M * Kc * T * N / t / 1e9, (M * Kc * T + M * N) * 4 / t / 1e9), flush=True)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    for mode in ("real", "alias"):
        for variant in (0, 3):
            run(M, 64, 64, 5, mode, variant)
