"""Forward time of the one-launch small-graph kernels against the filter order K (set-up vs per-step cost). Developer tool."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tgcn_amd import functional as F, _lib
dev = torch.device("cuda:0")
for wl, q, C, N in (("cfg2", 128, 1, 64), ("cfg3", 64, 28, 64)):
    op, spec = bench.build_workload(wl, "random", dev)
    x = torch.randn(q, op.n, C, device=dev)
    for K in (1, 2, 3, 5, 9):
        W = torch.randn(K, C, N, device=dev) * 0.1
        fold = F.power_fold_matrix(K, dev) if K > 2 else None
        for _ in range(5): F.cheb_forward_small(op, x, W, fold, None, 0, 0)
        torch.cuda.synchronize()
        _lib.profile_start(64)
        for _ in range(20): F.cheb_forward_small(op, x, W, fold, None, 0, 0)
        torch.cuda.synchronize()
        pr = _lib.profile_stop(64)
        print("%s K=%d: %.1f us" % (wl, K, 1e3 * sum(ms for _, ms in pr) / len(pr)), flush=True)
