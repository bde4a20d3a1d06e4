"""A model written the way the reference's training scripts write theirs (two Chebyshev layers, ReLU, gcn_pool_4,
linear head; cf. examples/pytorch_based/pytorch_hcp_tgcn.py:93-155), importing the layers through the compat path,
trained for a few SGD steps on synthetic data.  GPU only."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as TF

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _ring_graph(n, extra, rng):
    """symmetric normalised operand -D^-1/2 A D^-1/2 of a ring with a few chords, as a dense tensor (what the scripts pass)."""
    A = np.zeros((n, n), np.float32)
    for i in range(n):
        for d in (1, 2):
            A[i, (i + d) % n] = A[(i + d) % n, i] = 1.0
    for _ in range(extra):
        a, b = rng.integers(0, n, 2)
        if a != b:
            A[a, b] = A[b, a] = 1.0
    dis = 1.0 / np.sqrt(A.sum(0))
    return torch.tensor(-(dis[:, None] * A * dis[None, :]), dtype=torch.float32)


def test_two_layer_model_trains(gpu_device):
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        from tgcn.nn.gcn import GCNCheb, TGCNCheb_H, gcn_pool_4          # the reference's import line, unchanged
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(0)
    L0, L2 = _ring_graph(160, 30, rng), _ring_graph(40, 8, rng)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.tgcn1 = TGCNCheb_H(L0, 1, 16, 6, 15)
            self.gcn2 = GCNCheb(L2, 16, 24, 5)
            self.fc = nn.Linear(10 * 24, 6)

        def forward(self, x):
            x = gcn_pool_4(TF.relu(self.tgcn1(x)))
            x = gcn_pool_4(TF.relu(self.gcn2(x)))
            return TF.log_softmax(self.fc(x.view(x.shape[0], -1)), dim=1)

    torch.manual_seed(0)
    net = Net().cuda()
    assert sorted(net.state_dict()) == ["fc.bias", "fc.weight", "gcn2.bias", "gcn2.weight", "tgcn1.bias", "tgcn1.weight"]
    x = torch.randn(64, 160, 15, device="cuda")
    y = (x[:, :40].mean(dim=(1, 2)) > 0).long() + 2 * (x[:, 80:120].mean(dim=(1, 2)) > 0).long()   # 4 learnable classes
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = TF.nll_loss(net(x), y)
        loss.backward()
        for p in net.parameters():
            assert torch.isfinite(p.grad).all()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    # a checkpoint round trip keeps the function (L is a plain attribute and is not part of the state dict)
    net2 = Net().cuda()
    net2.load_state_dict(net.state_dict())
    with torch.no_grad():
        assert torch.equal(net(x), net2(x))


def test_training_step_is_hipgraph_capturable(gpu_device):
    """Forward + backward + SGD of a two-layer model captured into ONE hipGraph (torch.cuda.graphs) and replayed on new
    batches: same losses as the eager loop.  Small-batch steps are bound by the host (autograd engine, ~0.2 ms); a
    captured step issues in a few microseconds."""
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        from tgcn.nn.gcn import GCNCheb, TGCNCheb_H, gcn_pool_4
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(1)
    L0, L2 = _ring_graph(160, 30, rng), _ring_graph(40, 8, rng)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.tgcn1 = TGCNCheb_H(L0, 1, 16, 6, 15)
            self.gcn2 = GCNCheb(L2, 16, 24, 5)
            self.fc = nn.Linear(10 * 24, 6)

        def forward(self, x):
            x = gcn_pool_4(TF.relu(self.tgcn1(x)))
            x = gcn_pool_4(TF.relu(self.gcn2(x)))
            return TF.log_softmax(self.fc(x.view(x.shape[0], -1)), dim=1)

    def make():
        torch.manual_seed(3)
        net = Net().cuda()
        return net, torch.optim.SGD(net.parameters(), lr=0.05)

    batches = [(torch.randn(32, 160, 15, device="cuda"), torch.randint(0, 6, (32,), device="cuda")) for _ in range(5)]
    # eager reference
    net, opt = make()
    eager = []
    for x, y in batches:
        opt.zero_grad(set_to_none=True)
        loss = TF.nll_loss(net(x), y)
        loss.backward()
        opt.step()
        eager.append(float(loss.detach()))
    # captured: warm-up steps on a side stream (the documented torch.cuda.graphs recipe), then capture one step
    net, opt = make()
    xs, ys = batches[0][0].clone(), batches[0][1].clone()
    snapshot = [p.detach().clone() for p in net.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            TF.nll_loss(net(xs), ys).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad():
        for p, p0 in zip(net.parameters(), snapshot):      # undo the warm-up updates
            p.copy_(p0)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss_static = TF.nll_loss(net(xs), ys)
        loss_static.backward()
        opt.step()
    with torch.no_grad():
        for p, p0 in zip(net.parameters(), snapshot):      # the capture itself does not run, but be explicit
            p.copy_(p0)
    got = []
    for x, y in batches:
        xs.copy_(x)
        ys.copy_(y)
        g.replay()
        got.append(float(loss_static.detach()))
    assert np.allclose(got, eager, rtol=2e-4, atol=1e-5), (got, eager)


@pytest.mark.parametrize("workload", ["cfg3", "cfg4", "cfg2w"])
def test_bench_line_of_the_other_workloads(workload, gpu_device):
    """`python bench.py --workload W` end to end for BASELINE.json's other configurations: one JSON line with `roofline` and a `cpu_baseline` whose
    time-step loop also works when a time step takes microseconds (round 6: the budget loop divided by a rounded-to-zero duration on cfg3)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "10", "--warmup", "3"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["roofline"]["frac"] > 0 and "other_workloads" not in line
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["gpu_vs_cpu_rel_err"] <= 1e-5 and cpu["scaled_from"] is None and cpu["host"]["os_cpu_count"] >= 1
    assert len(cpu["samples_timed"]) == {"cfg3": 64, "cfg4": 1, "cfg2w": 128}[workload]
    if workload != "cfg4":
        assert cpu["torch_dense_einsum"]["gpu_vs_cpu_rel_err"] <= 1e-5
