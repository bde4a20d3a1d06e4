#!/usr/bin/env python3
"""Host-side latency of the small-graph configs: forwards per second when launches are issued back to back
(no sync between calls) and a breakdown of where the host time goes.  Developer tool."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for wl in ("cfg3", "cfg2"):
        op, spec = bench.build_workload(wl, "random", dev)
        layer = bench.make_layer(op, spec, dev)
        x = bench.make_input(op, spec, dev, 0)
        with torch.no_grad():
            for _ in range(20):
                layer(x)
            torch.cuda.synchronize()
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                layer(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("%s: host issue %.1f us/forward, incl. drain %.1f us/forward" % (wl, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        with torch.no_grad():
            pr.enable()
            for _ in range(200):
                layer(x)
            pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()
