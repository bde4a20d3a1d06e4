"""Host issue time of one forward + backward step of a layer, next to a plain torch matmul of similar size (the autograd
engine, not the layer, sets the floor on small batches). Developer tool."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
dev = torch.device("cuda:0")
for wl in ("cfg3", "cfg2"):
    op, spec = bench.build_workload(wl, "random", dev)
    layer = bench.make_layer(op, spec, dev)
    x = bench.make_input(op, spec, dev, 0)
    g = torch.randn_like(layer(x))
    def step():
        out = layer(x); out.backward(g); layer.zero_grad(set_to_none=True)
    for _ in range(30): step()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(wl, "issue %.1f us/step, with drain %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6), flush=True)
    # a plain torch op pair of similar size for reference
    a = torch.randn(64, 784, 64, device=dev, requires_grad=True)
    w = torch.randn(64, 64, device=dev, requires_grad=True)
    def tstep():
        o = a @ w; o.backward(g[:64] if g.shape[0] >= 64 else torch.ones_like(o)); a.grad = None; w.grad = None
    for _ in range(30): tstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tstep()
    torch.cuda.synchronize()
    print("   torch matmul fwd+bwd of a similar size: %.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
