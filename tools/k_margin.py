#!/usr/bin/env python3
"""Error of the folded-monomial layer against float64 as the filter order grows (how thin is the margin at K = 25?):
GCNCheb(L, 64, 64, K) on an 8,281-vertex mesh, 2 samples (16,562 rows: the bf16x3 projection is the shipped choice), K = 5 ... 64, per projection
variant (0 shipped, 3 bf16x3 forced, 4 exact fp32), next to the error of the reference's OWN evaluation order in fp32 (numpy).  Developer tool."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import tgcn_amd
    from tgcn_amd import _lib
    from tools import synth
    n, row, col, val = synth.sheet_mesh(91)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val)
    L32 = op.to_scipy().astype(np.float32)
    L = L32.astype(np.float64)
    rng = np.random.default_rng(25)
    x = rng.standard_normal((2, n, 64)).astype(np.float32)
    for K in (5, 10, 25, 32, 48, 64):
        torch.manual_seed(5)
        layer = tgcn_amd.GCNCheb(op, 64, 64, K).cuda()
        W = layer.weight.detach().double().cpu().numpy()
        b = layer.bias.detach().double().cpu().numpy()
        Xt, P = [x.astype(np.float64)], x.astype(np.float64)
        Xs, Ps = [x], x
        for k in range(1, K):
            P = np.stack([L @ P[i] for i in range(2)])
            Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
            Ps = np.stack([L32 @ Ps[i] for i in range(2)]).astype(np.float32)
            Xs.append(Ps if k == 1 else (2 * Ps - Xs[k - 2]).astype(np.float32))
        ref = sum(Xt[k] @ W[k] for k in range(K)) + b
        ref32 = (sum((Xs[k] @ W[k].astype(np.float32)) for k in range(K)) + b.astype(np.float32)).astype(np.float32)
        scale = np.abs(ref).max()
        res = dict(K=K, reference_order_fp32=float(np.abs(ref32 - ref).max() / scale))
        for variant in (0, 3, 4):
            _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", variant))
            with torch.no_grad():
                out = layer(torch.as_tensor(x).cuda())
            res["variant_%d" % variant] = float(np.abs(out.cpu().numpy() - ref).max() / scale)
        _lib.lib().tgcn_reset_tuning()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
