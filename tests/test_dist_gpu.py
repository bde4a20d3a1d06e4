"""N = 2 on ONE GPU (gloo rendezvous, both ranks on cuda:0): the vertex-sharded layer with its real HIP compute
(the CPU suite runs the same communication code with the oracle injected, tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, exchange, banded, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        torch.cuda.set_device(0)
        rng = np.random.default_rng(3)
        n, q, C, N, K = 3000, 2, 8, 12, 4
        if banded:
            row = np.repeat(np.arange(n), 6)
            col = np.clip(row + rng.integers(-9, 10, row.shape[0]), 0, n - 1)
        else:
            row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
        row = np.concatenate([row, np.full(500, 7)])
        col = np.concatenate([col, rng.integers(0, n, 500)])
        val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        dev = torch.device("cuda:0")
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
        args = (torch.as_tensor(x[:, sh.lo:sh.hi]).to(dev), torch.as_tensor(W).to(dev), torch.as_tensor(bias[sh.lo:sh.hi]).to(dev), 2, 1)
        out = sh.forward(*args)                                   # overlapped: HIP pack kernel, in-place receives, interior / boundary hops
        assert torch.equal(out, sh.forward(*args, overlap=False)), "overlapped and plain forms differ"
        if exchange == "halo":
            assert 0 < sh.n_int < sh.owned
        L = O.coo_to_csr(row, col, val, n)
        ref = np.einsum("kqnc,kcg->qng", O.stack_chebyshev(L, x, K).astype(np.float64), W.astype(np.float64)) + bias
        err = np.abs(out.cpu().numpy() - ref[:, sh.lo:sh.hi]).max() / np.abs(ref).max()
        ret[rank] = (float(err), sh.exchange, sh.owned)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,banded", [("halo", True), ("allgather", False)])
def test_vertex_sharded_hip_two_ranks_one_gpu(exchange, banded, gpu_device):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), exchange, banded, ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][2] for r in range(world)) == 3000
    for r in range(world):
        assert ret[r][0] <= 1e-5 and ret[r][1] == exchange, ret[r]


def _pf_worker(rank, world, port, exchange, banded, mode, C, N, ret):
    """round 6: project-first inside the shard and the backward through the transposed shard, HIP kernels, two ranks on one GPU"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(17)
        n, q, K = 3000, 3, 5
        if banded:
            row = np.repeat(np.arange(n), 6)
            col = np.clip(row + rng.integers(-9, 10, row.shape[0]), 0, n - 1)
        else:
            row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
        row = np.concatenate([row, np.full(500, 7)])
        col = np.concatenate([col, rng.integers(0, n, 500)])
        val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        g = rng.standard_normal((q, n, N)).astype(np.float32)
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
        xl = torch.as_tensor(x[:, sh.lo:sh.hi]).to(dev).contiguous()
        Wd, bl = torch.as_tensor(W).to(dev), torch.as_tensor(bias[sh.lo:sh.hi]).to(dev)
        out = sh.layer(xl, Wd, bl, 2, mode)
        assert sh.describe()["row_floats"] == (N if 2 * N <= C else C)
        assert torch.equal(out, sh.layer(xl, Wd, bl, 2, mode, overlap=False)), "overlapped and plain forms differ"
        assert torch.equal(out, sh.layer(xl, Wd, bl, 2, mode, depth=3))
        other = sh.layer(xl, Wd, bl, 2, mode, project_first=not sh.use_project_first(C, N, K))       # the other evaluation order
        gx, gW, gb = sh.layer_backward(xl, Wd, torch.as_tensor(g[:, sh.lo:sh.hi]).to(dev).contiguous(), 2, mode)
        L = O.coo_to_csr(row, col, val, n)
        stack = O.stack_reference_power if mode == 0 else O.stack_chebyshev
        ref = np.einsum("kqnc,kcg->qng", stack(L, x, K).astype(np.float64), W.astype(np.float64)) + bias
        rx, rW = O.layer_backward(L, x, W, g, "power" if mode == 0 else "chebyshev")
        rel = lambda a, b, full: float(np.abs(a - b).max() / np.abs(full).max())          # noqa: E731
        ret[rank] = (rel(out.cpu().numpy(), ref[:, sh.lo:sh.hi], ref), rel(other.cpu().numpy(), ref[:, sh.lo:sh.hi], ref),
                     rel(gx.cpu().numpy(), rx[:, sh.lo:sh.hi], rx), rel(gW.cpu().numpy(), rW, rW),
                     rel(gb.cpu().numpy(), g.astype(np.float64).sum(0)[sh.lo:sh.hi], g.astype(np.float64).sum(0)), sh.exchange, sh.owned)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,banded,mode,C,N", [("halo", True, 0, 64, 8), ("halo", True, 1, 64, 8), ("allgather", False, 0, 40, 12), ("halo", True, 1, 8, 12),
                                                      ("allgather", False, 1, 8, 12), ("halo", True, 0, 8, 32)])
def test_sharded_project_first_and_backward_hip_two_ranks_one_gpu(exchange, banded, mode, C, N, gpu_device):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pf_worker, args=(world, _free_port(), exchange, banded, mode, C, N, ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][6] for r in range(world)) == 3000
    for r in range(world):
        e_out, e_other, e_x, e_W, e_b, used, _ = ret[r]
        assert used == exchange and max(e_out, e_other) <= 1e-5 and max(e_x, e_W, e_b) <= 2e-5, ret[r]


def _cfg4_worker(rank, world, port, ret):
    """BASELINE.json configs[3] at FULL size, vertex-sharded over two ranks on one GPU (gloo transport, HIP kernels), through the module surface:
    sheet_mesh(300), ShardedTGCNCheb_H(L, 1, 32, 5, 1200), q = 1 -- against oracle/cheb_ref.c on the whole graph"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_port
        from tgcn_amd import dist as tdist
        from tools import synth
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        n, row, col, val = synth.sheet_mesh(300, device=dev)
        torch.manual_seed(1)
        layer = tdist.ShardedTGCNCheb_H(tdist.CooGraph(n, row, col, val), 1, 32, 5, 1200, exchange="auto").to(dev)
        gen = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((1, n, 1200), device=dev, generator=gen)
        lo, hi = layer.owned_rows(dev)
        sh = layer.shard(dev)
        xl = x[:, lo:hi].contiguous()
        with torch.no_grad():
            out = layer(xl)
            layer.overlap = False
            plain = layer(xl)
        d = sh.describe()
        same = bool(torch.equal(out, plain))
        order = torch.argsort(row * n + col, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
        ref = c_port.forward(0, rowptr.to(torch.int32).cpu().numpy(), col[order].to(torch.int32).cpu().numpy(), val[order].cpu().numpy(),
                             x.cpu().numpy(), layer.weight.detach().reshape(5, 1200, 32).cpu().numpy(), layer.bias.detach().reshape(-1).cpu().numpy(), 2)
        err = float(np.abs(out.cpu().numpy() - ref[:, lo:hi]).max() / np.abs(ref).max())
        ret[rank] = (err, same, sh.exchange, d["row_floats"], d["bytes_in_per_hop_and_time_step"], d["rows_in_per_hop"], hi - lo)
    finally:
        dist.destroy_process_group()


def test_cfg4_full_size_vertex_sharded_world2_one_gpu(gpu_device):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cfg4_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][6] for r in range(world)) == 90000
    for r in range(world):
        err, same, used, width, nbytes, rows_in, _ = ret[r]
        assert err <= 1e-5 and same, ret[r]
        assert used == "halo" and width == 32 and nbytes == rows_in * 32 * 4        # 32-float halo rows, not 1200-float ones


# ---------------------------------------------------------------------------------------------------------------------------------
# First contact with RCCL.  A one-GPU box cannot hold two NCCL ranks (RCCL refuses two ranks on one device), so
# the N > 1 code meets the real backend at world size 1: init_process_group("nccl", device_id=...), all_reduce on a device tensor,
# all_gather_into_tensor(async_op=True) + work.wait() with RCCL's stream ordering (the collective runs on RCCL's own stream; wait() only
# makes torch's current stream wait for it -- gloo blocks the host instead), batch_isend_irecv-free halo form, and bench.py's N > 1 path.
def _nccl_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        t = torch.tensor([3.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 3.5
        rng = np.random.default_rng(5)
        n, q, C, N, K = 5000, 5, 64, 32, 5            # q = 5 with depth 2: full and ragged pipeline groups
        row, col = rng.integers(0, n, 10 * n), rng.integers(0, n, 10 * n)
        row = np.concatenate([row, np.full(700, 11)])               # one long row: segment + fix-up path inside the sharded hop
        col = np.concatenate([col, rng.integers(0, n, 700)])
        val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        L = O.coo_to_csr(row, col, val, n)
        res = {}
        for exchange in ("allgather", "halo"):
            for mode in (0, 1):
                sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
                assert sh.exchange == exchange and sh.owned == n
                args = (torch.as_tensor(x).to(dev), torch.as_tensor(W).to(dev), torch.as_tensor(bias).to(dev), 2, mode)
                out = sh.forward(*args)                       # overlapped: async all-gather / p2p batch, wait() on the compute stream
                out3 = sh.forward(*args, depth=3)
                plain = sh.forward(*args, overlap=False)
                torch.cuda.synchronize()
                assert torch.equal(out, plain) and torch.equal(out3, plain), "overlapped and plain forms differ under RCCL"
                if mode == 1:
                    stack = O.stack_chebyshev(L, x, K)
                else:           # mode 0 takes the weight in the monomial basis: terms L^k x
                    stack = np.empty((K,) + x.shape, np.float32)
                    stack[0] = x
                    for k in range(1, K):
                        stack[k] = O._apply(L, stack[k - 1])
                ref = np.einsum("kqnc,kcg->qng", stack.astype(np.float64), W.astype(np.float64)) + bias
                res[(exchange, mode)] = float(np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max())
        ret["err"] = res
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_first_contact_with_rccl_world_1(gpu_device):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret["backend"] == "nccl"
    assert len(ret["err"]) == 4 and all(e <= 1e-5 for e in ret["err"].values()), dict(ret["err"])


def test_bench_n_gt_1_path_on_rccl_one_rank(gpu_device):
    """bench.py under the driver's own launcher with the nccl backend on one rank: rendezvous with device_id, barrier, the all-reduce of the
    timing, and the vertex-sharded extras (all-gather form, plain + overlapped) -- the code an 8-GPU lease would run, minus the peers."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "nccl", "--steps", "2", "--warmup", "1", "--no-cpu", "--vertices", "200000",
           "--entries", "3000000", "--force-extras", "--extras-exchange", "allgather", "--extras-budget", "120"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 1 and line["config"]["dist_backend"] == "nccl" and line["value"] > 0 and "extras_abandoned" not in line
    forms = [(e["shard"], e["form"], e.get("exchange"), "error" in e) for e in line["other_shardings"]]
    assert forms == [("vertex", "plain", "allgather", False), ("vertex", "overlapped", "allgather", False)], line["other_shardings"]
    for e in line["other_shardings"]:
        assert e["value"] > 0 and e["ranks"][0]["owned_rows"] == 200000 and e["ranks"][0]["phases_ms"]


def _nccl_peers_worker(rank, world, port, exchange, banded, ret):
    """the two-rank test above with REAL peers: rank r on cuda:r, RCCL transport (runs only where the box has the GPUs)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        rng = np.random.default_rng(3)
        n, q, C, N, K = 3000, 3, 8, 12, 4
        if banded:
            row = np.repeat(np.arange(n), 6)
            col = np.clip(row + rng.integers(-9, 10, row.shape[0]), 0, n - 1)
        else:
            row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
        val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
        args = (torch.as_tensor(x[:, sh.lo:sh.hi]).to(dev), torch.as_tensor(W).to(dev), torch.as_tensor(bias[sh.lo:sh.hi]).to(dev), 2, 1)
        out = sh.forward(*args)
        plain = sh.forward(*args, overlap=False)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, plain))
        L = O.coo_to_csr(row, col, val, n)
        ref = np.einsum("kqnc,kcg->qng", O.stack_chebyshev(L, x, K).astype(np.float64), W.astype(np.float64)) + bias
        err = np.abs(out.cpu().numpy() - ref[:, sh.lo:sh.hi]).max() / np.abs(ref).max()
        ret[rank] = (float(err), sh.exchange, sh.owned, same)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between real peers)")
@pytest.mark.parametrize("exchange,banded", [("halo", True), ("allgather", False)])
def test_vertex_sharded_two_ranks_two_gpus_rccl(exchange, banded):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_peers_worker, args=(world, _free_port(), exchange, banded, ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][2] for r in range(world)) == 3000
    for r in range(world):
        assert ret[r][0] <= 1e-5 and ret[r][1] == exchange and ret[r][3], ret[r]


def test_bench_cfg4_shard_vertex_line_without_a_launcher(gpu_device):
    """`python bench.py --workload cfg4 --shard vertex`: the process forms a one-rank RCCL group itself, steps the sharded MODULE
    (project-first inside, 32-float hop rows) and checks its output against oracle/cheb_ref.c like the single-GPU line"""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg4", "--shard", "vertex", "--steps", "5", "--warmup", "2"], cwd=root,
                       env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 1 and line["config"]["dist_backend"] == "nccl" and line["config"]["sharding"].startswith("vertex rows across ranks")
    assert line["roofline"]["path"].startswith("project-first") and line["roofline"]["launches_per_step"] == 4
    assert line["cpu_baseline"]["gpu_vs_cpu_rel_err"] <= 1e-5


def _edge_module_worker(rank, world, port, timed, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd import dist as tdist
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(41)
        n, E, q, f, g_ch, K, H = 4000, 30000, 2, 8, 6, 4, 5
        ei = rng.integers(0, n, (2, E))
        ei[1, : E // 10] = ei[0, : E // 10]
        w = rng.uniform(0.5, 1.5, E).astype(np.float32)
        edge_index, edge_weight = torch.as_tensor(ei).to(dev), torch.as_tensor(w).to(dev)
        torch.manual_seed(9)
        if timed:
            mod = tdist.ShardedChebTimeConv(1, g_ch, K, H).to(dev)
            x = rng.standard_normal((q, n, H)).astype(np.float32)
            fwd = O.cheb_time_conv_forward
        else:
            mod = tdist.ShardedChebConv(f, g_ch, K).to(dev)
            x = rng.standard_normal((q, n, f)).astype(np.float32)
            fwd = O.cheb_conv_forward
        lo, hi = mod.owned_rows(dev, edge_index, n, edge_weight)
        xl = torch.as_tensor(np.ascontiguousarray(x[:, lo:hi])).to(dev).requires_grad_(True)
        out = mod(xl, edge_index, edge_weight)
        gout = rng.standard_normal((q, n, g_ch)).astype(np.float32)
        out.backward(torch.as_tensor(np.ascontiguousarray(gout[:, lo:hi])).to(dev))
        W, b = mod.weight.detach().cpu().numpy(), mod.bias.detach().cpu().numpy()
        ref = fwd(x, ei, w, W, b)
        r_, c_, lap = O.edge_laplacian(ei, w, n)
        xr = x if not timed else x[..., None]
        rx, rW = O.layer_backward(O.coo_to_csr(r_, c_, lap, n), xr, W, gout, "chebyshev")
        rx = rx.reshape(x.shape)
        rel = lambda a, b_, full: float(np.abs(a - b_).max() / np.abs(full).max())          # noqa: E731
        ret[rank] = (rel(out.detach().cpu().numpy(), ref[:, lo:hi], ref), rel(xl.grad.cpu().numpy(), rx[:, lo:hi], rx),
                     rel(mod.weight.grad.cpu().numpy(), rW, rW), rel(mod.bias.grad.cpu().numpy(), gout.astype(np.float64).sum((0, 1)), gout.astype(np.float64).sum((0, 1))), hi - lo)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("timed", [False, True])
def test_sharded_edge_list_modules_hip_two_ranks_one_gpu(timed, gpu_device):
    """ShardedChebConv / ShardedChebTimeConv with the library's edge normalisation and the HIP kernels, two ranks on one GPU"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_edge_module_worker, args=(world, _free_port(), timed, ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][4] for r in range(world)) == 4000
    for r in range(world):
        assert ret[r][0] <= 1e-5 and max(ret[r][1:4]) <= 2e-5, ret[r]


def _fuzz_worker(rank, world, port, seed, cases, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(seed)
        out = []
        for case in range(cases):
            n = int(rng.integers(300, 5000))
            q = int(rng.integers(1, 4))
            K = int(rng.choice([1, 2, 3, 5]))
            C = int(rng.choice([1, 3, 4, 7, 16, 33, 64, 100]))
            N = int(rng.choice([1, 2, 5, 8, 17, 32]))
            mode = int(rng.integers(0, 2))
            bias_kind = int(rng.integers(0, 3))
            exchange = str(rng.choice(["halo", "allgather"]))
            if exchange == "halo":
                row = np.repeat(np.arange(n), 5)
                col = np.clip(row + rng.integers(-12, 13, row.shape[0]), 0, n - 1)
            else:
                row, col = rng.integers(0, n, 6 * n), rng.integers(0, n, 6 * n)
            row = np.concatenate([row, np.full(150, 3)])                # one long row: segments + fix-up inside the sharded hop
            col = np.concatenate([col, rng.integers(0, n, 150)])
            val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
            x = rng.standard_normal((q, n, C)).astype(np.float32)
            W = (rng.standard_normal((K, C, N)) / 3).astype(np.float32)
            bias = None if bias_kind == 0 else (rng.standard_normal(N).astype(np.float32) if bias_kind == 1 else rng.standard_normal((n, N)).astype(np.float32))
            g = rng.standard_normal((q, n, N)).astype(np.float32)
            sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
            xl = torch.as_tensor(np.ascontiguousarray(x[:, sh.lo:sh.hi])).to(dev)
            bl = None if bias is None else torch.as_tensor(np.ascontiguousarray(bias if bias_kind == 1 else bias[sh.lo:sh.hi])).to(dev)
            Wt = torch.as_tensor(W).to(dev)
            o1 = sh.layer(xl, Wt, bl, bias_kind, mode)
            same = bool(torch.equal(o1, sh.layer(xl, Wt, bl, bias_kind, mode, overlap=False)))
            o2 = sh.layer(xl, Wt, bl, bias_kind, mode, project_first=not sh.use_project_first(C, N, K))
            gx, gW, gb = sh.layer_backward(xl, Wt, torch.as_tensor(np.ascontiguousarray(g[:, sh.lo:sh.hi])).to(dev), bias_kind, mode)
            L = O.coo_to_csr(row, col, val, n)
            stack = O.stack_reference_power if mode == 0 else O.stack_chebyshev
            ref = np.einsum("kqnc,kcg->qng", stack(L, x, K).astype(np.float64), W.astype(np.float64))
            if bias is not None:
                ref = ref + bias
            rx, rW = O.layer_backward(L, x, W, g, "power" if mode == 0 else "chebyshev")
            s_o, s_x, s_w = max(np.abs(ref).max(), 1e-30), max(np.abs(rx).max(), 1e-30), max(np.abs(rW).max(), 1e-30)
            errs = [float(np.abs(o1.cpu().numpy() - ref[:, sh.lo:sh.hi]).max() / s_o), float(np.abs(o2.cpu().numpy() - ref[:, sh.lo:sh.hi]).max() / s_o),
                    float(np.abs(gx.cpu().numpy() - rx[:, sh.lo:sh.hi]).max() / s_x), float(np.abs(gW.cpu().numpy() - rW).max() / s_w)]
            out.append((dict(n=n, q=q, K=K, C=C, N=N, mode=mode, bias_kind=bias_kind, exchange=exchange), same, errs))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_random_sweep_of_the_sharded_layer_hip_two_ranks_one_gpu(gpu_device):
    """16 random shapes with the HIP kernels (odd widths: unaligned hop / projection forms under the sharded engine; a long row per graph; both
    evaluation orders; both recurrences; three bias kinds): overlapped == plain bit for bit, outputs and gradients against the oracle"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fuzz_worker, args=(world, _free_port(), 515, 16, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for cfg, same, errs in ret[r]:
            assert same, cfg
            assert max(errs[:2]) <= 1e-5 and max(errs[2:]) <= 5e-5, (cfg, errs)


def _nccl_module_worker(rank, world, port, ret):
    """the sharded MODULE between real peers: rank r on cuda:r, RCCL transport (runs only where the box has the GPUs)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from tgcn_amd import dist as tdist
        from tools import synth
        n, row, col, val = synth.sheet_mesh(100, device="cpu")            # the CPU generator: the same graph on every rank whatever its device
        H, g_ch, K, q = 240, 32, 5, 2
        torch.manual_seed(1)
        layer = tdist.ShardedTGCNCheb_H(tdist.CooGraph(n, row, col, val), 1, g_ch, K, H).to(dev)
        rng = np.random.default_rng(5)
        x = rng.standard_normal((q, n, H)).astype(np.float32)
        gout = rng.standard_normal((q, n, g_ch)).astype(np.float32)
        lo, hi = layer.owned_rows(dev)
        xl = torch.as_tensor(np.ascontiguousarray(x[:, lo:hi])).to(dev).requires_grad_(True)
        out = layer(xl)
        layer.overlap = False
        with torch.no_grad():
            plain = layer(xl)
        layer.overlap = True
        out.backward(torch.as_tensor(np.ascontiguousarray(gout[:, lo:hi])).to(dev))
        torch.cuda.synchronize()
        L = O.coo_to_csr(row.numpy(), col.numpy(), val.numpy(), n)
        W, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
        ref = O.tgcn_cheb_h_forward(L, x, W, b)
        rx, rW = O.layer_backward(L, x[..., None], W, gout, "power")
        rel = lambda a, b_, full: float(np.abs(a - b_).max() / np.abs(full).max())          # noqa: E731
        ret[rank] = (rel(out.detach().cpu().numpy(), ref[:, lo:hi], ref), bool(torch.equal(out.detach(), plain)),
                     rel(xl.grad.cpu().numpy(), rx.reshape(x.shape)[:, lo:hi], rx), rel(layer.weight.grad.cpu().numpy(), rW, rW),
                     rel(layer.bias.grad.cpu().numpy(), gout.astype(np.float64).sum(0)[None], gout.astype(np.float64).sum(0)), layer.shard(dev).describe()["row_floats"], hi - lo)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between real peers)")
def test_sharded_module_two_ranks_two_gpus_rccl():
    """ShardedTGCNCheb_H (project-first: 32-float halo rows) forward + backward over RCCL between two GPUs: never run on this pool (one GPU per lease),
    kept ready for the first box that has a peer -- the same assertions as the two-ranks-one-GPU gloo tests above"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_module_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][6] for r in range(world)) == 10000
    for r in range(world):
        e_out, same, e_x, e_W, e_b, width, _ = ret[r]
        assert e_out <= 1e-5 and same and max(e_x, e_W, e_b) <= 2e-5 and width == 32, ret[r]


def _cfg5_worker(rank, world, port, ret):
    """BASELINE.json configs[4] at FULL size, vertex-sharded over two ranks on one GPU (gloo transport, HIP kernels): R-MAT 10 M / 160 M,
    ShardedTGCNCheb(L, 64, 64, 5), ONE time step, against oracle/cheb_ref.c on the whole graph"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="64")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_port
        from tgcn_amd import dist as tdist
        from tools import synth
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        n, nnz = 10_000_000, 160_000_000
        _, row, col, val = synth.rmat(n, nnz, seed=12345, labeling="random", device=dev)
        torch.manual_seed(1)
        layer = tdist.ShardedTGCNCheb(tdist.CooGraph(n, row, col, val), 64, 64, 5, exchange="auto").to(dev)
        lo, hi = layer.owned_rows(dev)
        sh = layer.shard(dev)
        layer.L = tdist.CooGraph(n, row[:0], col[:0], val[:0])
        gen = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((1, n, 64), device=dev, generator=gen)
        xl = x[:, lo:hi].contiguous()
        with torch.no_grad():
            out = layer(xl)
        torch.cuda.synchronize()
        d = sh.describe()
        # the whole-graph oracle on this rank's host cores (both ranks compute it: ~2 x 12 s sharing the box)
        order = torch.argsort(row * n + col)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
        rp, ci, va = rowptr.to(torch.int32).cpu().numpy(), col[order].to(torch.int32).cpu().numpy(), val[order].cpu().numpy()
        del row, col, val, order
        ref = c_port.forward(0, rp, ci, va, x.cpu().numpy(), layer.weight.detach().cpu().numpy(), layer.bias.detach().reshape(-1).cpu().numpy(), 2)
        err = float(np.abs(out.cpu().numpy() - ref[:, lo:hi]).max() / np.abs(ref).max())
        ret[rank] = (err, sh.exchange, d["row_floats"], d["halo_rows"], d["bytes_in_per_hop_and_time_step"], hi - lo, sh.n_int)
    finally:
        dist.destroy_process_group()


def test_cfg5_full_size_vertex_sharded_world2_one_gpu(gpu_device):
    """the headline graph through the vertex-sharded module at FULL size: 5 M-row shards, ~1.9 M halo rows of 64 floats per hop (the halo form: 38 % of
    the remote vertices), interior + boundary operands with long rows and fix-ups, 64-bit offsets -- <= 1e-5 against the C restatement"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cfg5_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][5] for r in range(world)) == 10_000_000
    for r in range(world):
        err, used, width, halo, nbytes, owned, n_int = ret[r]
        assert err <= 1e-5, ret[r]
        assert used == "halo" and width == 64 and 1_500_000 < halo < 2_500_000 and nbytes == halo * 256 and 0 < n_int < owned, ret[r]
