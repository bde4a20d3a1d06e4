#!/usr/bin/env python3
"""What the callers' gcn_pool_4(F.relu(layer(x))) (examples/pytorch_based/pytorch_hcp_tgcn.py:134-141) costs where the relu + pool epilogue is
NOT fused into the projection: shapes on the vertex-major layout 1 (short per-sample rows, C < 32: the HCP horizon) and
shapes with >= 96 output columns (project_x3v2_kernel).  Per shape: the layer alone, cheb_relu_pool (layer + one relu / pool pass over scratch),
and the pass by itself from the library's launch events -> one JSON line per shape.  Developer tool."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def main():
    import tgcn_amd
    from tgcn_amd import functional as F
    from tools import synth
    dev = torch.device("cuda:0")
    n, row, col, val = synth.sheet_mesh(244, device=dev)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val, dev)
    shapes = [("TGCNCheb_H", dict(f=1, g=32, K=10, H=15), 8, "layout 1 (C = 15 -> 16 floats per sample row): the reference's HCP layer at mesh size"),
              ("TGCNCheb_H", dict(f=1, g=64, K=25, H=15), 8, "layout 1, K = 25"),
              ("GCNCheb", dict(f=64, g=128, K=5), 8, ">= 96 output columns: project_x3v2_kernel"),
              ("GCNCheb", dict(f=32, g=64, K=5), 8, "fused reference point: <= 64 columns on layout 0")]
    for cls, a, q, note in shapes:
        torch.manual_seed(0)
        if cls == "TGCNCheb_H":
            layer = tgcn_amd.TGCNCheb_H(op, a["f"], a["g"], a["K"], a["H"]).to(dev)
            x = torch.randn(q, n, a["H"], device=dev)
            C_row = a["H"] * a["f"]
        else:
            layer = tgcn_amd.GCNCheb(op, a["f"], a["g"], a["K"]).to(dev)
            x = torch.randn(q, n, a["f"], device=dev)
            C_row = a["f"]
        Cp = C_row + (-C_row) % 4 if C_row >= 7 else C_row
        fused = bool(F.choose_layout(q, n, Cp) == 0 and F.pool_epilogue_is_fused(op, q, Cp, a["g"], a["K"], 4))
        res = {}
        with torch.no_grad():
            for name, fn in (("layer_ms", lambda: layer(x)), ("layer_relu_pool4_ms", lambda: tgcn_amd.cheb_relu_pool(layer, x, pool=4)),
                             ("pool_pass_alone_ms", None)):
                if fn is None:
                    y = layer(x)
                    fn = lambda: F.ReluPoolFn.apply(y, 4)          # noqa: E731
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    fn()
                torch.cuda.synchronize()
                res[name] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
        out_mb = q * n * a["g"] * 4 / 1e6
        print(json.dumps(dict(shape="%s(L,%s) q=%d on the %d-vertex mesh" % (cls, ",".join(str(v) for v in a.values()), q, n), note=note, epilogue_fused_into_projection=fused,
                              layer_output_MB=round(out_mb, 1), **res,
                              epilogue_cost_share=round((res["layer_relu_pool4_ms"] - res["layer_ms"]) / res["layer_relu_pool4_ms"], 4))), flush=True)


if __name__ == "__main__":
    main()
