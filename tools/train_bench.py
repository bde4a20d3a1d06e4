#!/usr/bin/env python3
"""Forward + backward time of the small-graph configs (the shapes the reference trains on), with a breakdown of
the kernels of one training step.  Developer tool.

    python tools/train_bench.py [--workloads cfg3,cfg2,hcp148] [--steps 100]
"""
import argparse
import collections
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402

KINDS = {0: "hop", 1: "fixup", 2: "project", 3: "relayout", 4: "small_fwd", 5: "wgrad", 6: "small_basis"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg3,cfg2,hcp148")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--no-dx", action="store_true", help="input does not require a gradient (first layer of a model)")
    args = ap.parse_args()
    from tgcn_amd import _lib
    dev = torch.device("cuda:0")
    for wl in args.workloads.split(","):
        op, spec = bench.build_workload(wl, "random", dev)
        layer = bench.make_layer(op, spec, dev)
        x = bench.make_input(op, spec, dev, 0).requires_grad_(not args.no_dx)
        g = torch.randn_like(layer(x))

        def step():
            out = layer(x)
            out.backward(g)
            layer.zero_grad(set_to_none=True)
            x.grad = None

        for _ in range(10):
            step()
        torch.cuda.synchronize()
        with torch.no_grad():
            t0 = time.perf_counter()
            for _ in range(args.steps):
                layer(x)
            torch.cuda.synchronize()
            tf = (time.perf_counter() - t0) / args.steps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        tfb = (time.perf_counter() - t0) / args.steps
        _lib.profile_start(4096)
        step()
        torch.cuda.synchronize()
        prof = _lib.profile_stop(4096)
        acc = collections.OrderedDict()
        for k, ms in prof:
            c = acc.setdefault(KINDS.get(k, str(k)), [0, 0.0])
            c[0] += 1
            c[1] += ms
        print("%s: forward %.1f us, forward+backward %.1f us; kernels of one step: %s" % (
            wl, tf * 1e6, tfb * 1e6, ", ".join("%s x%d %.1f us" % (k, c[0], c[1] * 1e3) for k, c in acc.items())), flush=True)
        print("    in launch order: " + " ".join("%s=%.1f" % (KINDS.get(k, str(k)), ms * 1e3) for k, ms in prof), flush=True)


if __name__ == "__main__":
    main()
