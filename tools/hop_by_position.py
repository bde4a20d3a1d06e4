#!/usr/bin/env python3
"""Per-position launch times of one cfg5 forward (developer tool): which of the K-1 hops of a time step costs what.
hop 1 gathers from x in the caller's labels (10 M rows), hops 2..K-1 from the compact hop tensors (4.73 M rows)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from tgcn_amd import _lib
    dev = torch.device("cuda:0")
    op, spec = bench.build_workload("cfg5", "random", dev)
    layer = bench.make_layer(op, spec, dev)
    x = bench.make_input(op, spec, dev, 0)
    with torch.no_grad():
        out = layer(x)
        out = None
        torch.cuda.synchronize()
        _lib.profile_start(4096)
        out = layer(x)
        torch.cuda.synchronize()
        prof = _lib.profile_stop(4096)
    hops = [ms for k, ms in prof if k == 0]
    fix = [ms for k, ms in prof if k == 1]
    K1 = spec["K"] - 1
    h = np.array(hops).reshape(-1, K1)
    f = np.array(fix).reshape(-1, K1)
    print("hop ms by position (mean over %d time steps):" % h.shape[0], np.round(h.mean(0), 3), " fix-up:", np.round(f.mean(0), 3))
    print("projection launches ms:", np.round([ms for k, ms in prof if k == 2], 3))


if __name__ == "__main__":
    main()
