"""Forward and forward+backward time of TGCNCheb_H layers on a 59.5 k-vertex sheet mesh (the size of the cortical mesh of
examples/pytorch_based/pygeo_hcp.py:461-462), general hops-then-projection path, with the kernel time by kind.  Developer tool."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, tgcn_amd
from tools import synth
from tgcn_amd import _lib
from tgcn_amd.graph import GraphOperand
dev = torch.device("cuda:0")
n, row, col, val = synth.sheet_mesh(244, device=dev)
op = GraphOperand.from_coo(n, row, col, val, dev)
for (K, H, g, q) in ((25, 15, 32, 8), (10, 15, 32, 32), (5, 1, 64, 64)):
    layer = tgcn_amd.TGCNCheb_H(op, 1, g, K, H).to(dev)
    x = torch.randn(q, n, H, device=dev, requires_grad=True)
    with torch.no_grad():
        for _ in range(5): layer(x)
        torch.cuda.synchronize()
        _lib.profile_start(4096)
        t0 = time.perf_counter()
        for _ in range(20): layer(x)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 20
        pr = _lib.profile_stop(4096)
    kinds = {}
    for k, ms in pr: kinds[k] = kinds.get(k, 0) + ms / 20
    go = torch.randn(q, n, g, device=dev)
    for _ in range(3):
        layer(x).backward(go); layer.zero_grad(); x.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        layer(x).backward(go); layer.zero_grad(); x.grad = None
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 10
    print("mesh n=%d nnz=%d TGCNCheb_H(L,1,%d,%d,%d) q=%d: forward %.3f ms (kernels by kind ms: %s), forward+backward %.3f ms" % (
        n, op.nnz, g, K, H, q, tf * 1e3, {k: round(v, 3) for k, v in kinds.items()}, tb * 1e3), flush=True)
