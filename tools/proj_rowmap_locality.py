#!/usr/bin/env python3
"""How much of the row-mapped projection's time is the scattered row maps? (developer experiment)
Times the two projection launches of the compacted cfg5 forward with the real row maps (compact rows = 47 % of the vertices, spread
evenly) and with fake contiguous ones (compact rows = the first n_c vertices, empty rows = the rest): same bytes, whole DRAM pages."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from tgcn_amd import _lib, functional as F
    dev = torch.device("cuda:0")
    op, spec = bench.build_workload("cfg5", "random", dev)
    layer = bench.make_layer(op, spec, dev)
    x = bench.make_input(op, spec, dev, 0)[:4].contiguous()
    plan = op.compact_plan()
    K = spec["K"]
    W = F.fold_weight(F.power_fold_matrix(K, dev), layer.weight.detach()).reshape(K * 64, 64).contiguous()
    bias = layer.bias.detach().reshape(op.n, 64).contiguous()
    real = (plan.rows, plan.empty)
    fake = (torch.arange(plan.n_c, dtype=torch.int32, device=dev), torch.arange(plan.n_c, op.n, dtype=torch.int32, device=dev))
    for name, (rows, empty) in (("real row maps", real), ("contiguous row maps", fake), ("real row maps", real)):
        plan.rows, plan.empty = rows, empty
        with torch.no_grad():
            F.cheb_forward_compact(plan, x, W, bias, 2, K, q_chunk=4)
            torch.cuda.synchronize()
            _lib.profile_start(4096)
            for _ in range(3):
                F.cheb_forward_compact(plan, x, W, bias, 2, K, q_chunk=4)
            torch.cuda.synchronize()
            prof = _lib.profile_stop(4096)
        pj = np.array([ms for k, ms in prof if k == 2]).reshape(-1, 2)
        print("%-20s projection of 4 time steps: compact rows %.3f ms, empty rows %.3f ms" % (name, pj[:, 0].mean(), pj[:, 1].mean()), flush=True)
    plan.rows, plan.empty = real


if __name__ == "__main__":
    main()
