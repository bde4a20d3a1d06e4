#!/usr/bin/env python3
"""One training step (forward, backward, SGD) of a model written like the reference's HCP network
(examples/pytorch_based/pytorch_hcp_tgcn.py:93-155: TGCNCheb_H(L0,1,32,10,15) -> relu -> gcn_pool_4 ->
GCNCheb(L2,32,64,10) -> relu -> gcn_pool_4 -> linear) at batch 512 on the 148-parcel graph, with the layers imported
through the compat path.  Developer tool: the number a user of the reference sees after switching.

    python tools/model_bench.py [--batch 512] [--steps 100] [--fused]
"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compat"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--fused", action="store_true", help="use tgcn_amd.cheb_relu_pool for layer + relu + pool")
    ap.add_argument("--graph", action="store_true", help="capture the training step into one hipGraph (torch.cuda.graphs) and replay it")
    args = ap.parse_args()
    from tgcn.nn.gcn import GCNCheb, TGCNCheb_H, gcn_pool_4           # the reference's import line
    import tgcn_amd
    z = np.load(os.path.join(ROOT, "tests", "golden", "TGCNChebH_dti148_q4_f1_g32_K10_H15.npz"))
    n = int(z["n"])
    import scipy.sparse as sp
    L0 = torch.tensor(sp.csr_matrix((z["val"], z["col"], z["rowptr"]), shape=(n, n)).toarray(), dtype=torch.float32)
    pad = (-n) % 16                                    # the reference's coarsening pads with isolated vertices (gcn/coarsening.py)
    L0 = TF.pad(L0, (0, pad, 0, pad))
    n += pad
    rng = np.random.default_rng(0)
    n2 = n // 4
    A = (rng.random((n2, n2)) < 0.5).astype(np.float32)
    A = np.triu(A, 1)
    A = A + A.T
    dis = 1.0 / np.sqrt(np.maximum(A.sum(0), 1))
    L2 = torch.tensor(-(dis[:, None] * A * dis[None, :]), dtype=torch.float32)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.tgcn1 = TGCNCheb_H(L0, 1, 32, 10, 15)
            self.gcn2 = GCNCheb(L2, 32, 64, 10)
            self.fc = nn.Linear((n2 // 4) * 64, 6)

        def forward(self, x):
            if args.fused:
                x = tgcn_amd.cheb_relu_pool(self.tgcn1, x, pool=4)
                x = tgcn_amd.cheb_relu_pool(self.gcn2, x, pool=4)
            else:
                x = gcn_pool_4(TF.relu(self.tgcn1(x)))
                x = gcn_pool_4(TF.relu(self.gcn2(x)))
            return TF.log_softmax(self.fc(x.reshape(x.shape[0], -1)), dim=1)

    torch.manual_seed(0)
    net = Net().cuda()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(args.batch, n, 15, device="cuda")
    y = torch.randint(0, 6, (args.batch,), device="cuda")

    def step():
        opt.zero_grad(set_to_none=True)
        loss = TF.nll_loss(net(x), y)
        loss.backward()
        opt.step()
        return loss

    if args.graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(5):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss = TF.nll_loss(net(x), y)
            loss.backward()
            opt.step()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    else:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t0) / args.steps
    print("HCP-style model, batch %d, %s: inference %.3f ms, training step %.3f ms (%.0f samples/s), loss %.4f" % (
        args.batch, ("fused relu+pool" if args.fused else "plain modules") + (", step replayed from a hipGraph" if args.graph else ""),
        df * 1e3, dt * 1e3, args.batch / dt, float(loss.detach())), flush=True)


if __name__ == "__main__":
    main()
