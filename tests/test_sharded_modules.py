"""Round 6: the vertex-sharded layer with the single-GPU path's drivers and the reference's module surface.

gloo, world size 2 / 3, on the CPU: the communication logic is the product code (tgcn_amd/dist.py: partition, tensor-only constructor,
halo / all-gather exchange, overlap, the transposed shard, gradient all-reduce); the arithmetic is the numpy stand-in of
tools/cpu_standins.py, checked against the oracle.  Reference call shapes: examples/pytorch_based/pytorch_hcp_tgcn.py:103-104 (TGCNCheb_H with a
wide horizon and a narrow output: project-first), :134 (forward), :167-169 (the examples train), :270-273 (model wrapped, caller unchanged).
tests/test_dist_gpu.py runs the same with the HIP kernels on one GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cheb_oracle as O
from tools.cpu_standins import CpuOps

TOL = 1e-5
GRAD_TOL = 2e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(n, seed, banded):
    """asymmetric pattern (L^T differs from L): the backward's transposed shard has its own halo lists"""
    rng = np.random.default_rng(seed)
    if banded:
        row = np.repeat(np.arange(n), 5)
        col = np.clip(row + rng.integers(-6, 7, row.shape[0]), 0, n - 1)
        extra = rng.integers(0, n, (2, n // 20))
        row, col = np.concatenate([row, extra[0]]), np.concatenate([col, extra[1]])
    else:
        row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
    row = np.concatenate([row, np.full(200, 7)])
    col = np.concatenate([col, rng.integers(0, n, 200)])
    val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
    return row, col, val


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def _ref_forward(L, x, W_ref, bias, mode):
    """oracle: out = sum_k T_k x W_k + bias in the REFERENCE's basis (mode 0: the dense-L classes' recursion)"""
    stack = O.stack_reference_power if mode == 0 else O.stack_chebyshev
    out = np.einsum("kqnc,kcg->qng", stack(L, x, W_ref.shape[0]).astype(np.float64), W_ref.astype(np.float64))
    return out if bias is None else out + bias


def _spawn(fn, world, *args):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(world, _free_port(), ret) + args, nprocs=world, join=True)
    assert len(ret) == world
    return [ret[r] for r in range(world)]


# ------------------------------------------------------------------------------------------------ project-first inside the shard
def _pf_worker(rank, world, port, ret, exchange, banded, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        n, q, C, N, K = 360, 3, 24, 4, 5               # 2 N <= C: project first -- hops and messages on N-wide rows
        row, col, val = _graph(n, 3, banded)
        rng = np.random.default_rng(4)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)            # reference basis
        bias = rng.standard_normal((n, N)).astype(np.float32)
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device="cpu", exchange=exchange, ops=CpuOps())
        xl, Wt, bl = torch.from_numpy(x[:, sh.lo:sh.hi]), torch.from_numpy(W), torch.from_numpy(bias[sh.lo:sh.hi])
        assert sh.use_project_first(C, N, K)
        out = sh.layer(xl, Wt, bl, 2, mode)                                    # raw weight: the fold happens inside
        d = sh.describe()
        assert d["row_floats"] == N and d["bytes_in_per_hop_and_time_step"] == d["rows_in_per_hop"] * N * 4      # rows x C_out x 4 bytes
        if sh.exchange == "halo":
            assert d["message_bytes_per_peer_in"] == [c * N * 4 for c in sh.recv_counts]
        assert torch.equal(out, sh.layer(xl, Wt, bl, 2, mode, overlap=False)), "overlapped and plain project-first forms differ"
        assert torch.equal(out, sh.layer(xl, Wt, bl, 2, mode, depth=3))
        hops_first = sh.layer(xl, Wt, bl, 2, mode, project_first=False)        # the other evaluation order on C-wide rows
        assert sh.describe()["row_floats"] == C
        L = O.coo_to_csr(row, col, val, n)
        ref = _ref_forward(L, x, W, bias, mode)[:, sh.lo:sh.hi]
        scale = np.abs(_ref_forward(L, x, W, bias, mode)).max()
        per_channel = sh.layer(xl, Wt, torch.from_numpy(bias[0]), 1, mode)       # bias kind 1 rides on Z_0 too
        ref1 = _ref_forward(L, x, W, bias[0], mode)[:, sh.lo:sh.hi]
        nobias = sh.layer(xl, Wt, None, 0, mode)
        ref0 = _ref_forward(L, x, W, None, mode)[:, sh.lo:sh.hi]
        ret[rank] = (float(np.abs(out.numpy() - ref).max() / scale), float(np.abs(hops_first.numpy() - ref).max() / scale),
                     float(np.abs(per_channel.numpy() - ref1).max() / scale), float(np.abs(nobias.numpy() - ref0).max() / scale), sh.exchange, sh.n_int, sh.owned)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,banded,mode", [(2, "halo", True, 0), (3, "halo", True, 1), (2, "allgather", False, 0), (3, "auto", False, 1)])
def test_project_first_inside_the_shard(world, exchange, banded, mode):
    res = _spawn(_pf_worker, world, exchange, banded, mode)
    assert sum(r[6] for r in res) == 360
    for e_pf, e_hf, e_ch, e_nb, used, n_int, owned in res:
        assert max(e_pf, e_hf, e_ch, e_nb) <= TOL, (e_pf, e_hf, e_ch, e_nb)
        if banded and used == "halo":
            assert 0 < n_int < owned


# ------------------------------------------------------------------------------------------------ gradients
def _grad_worker(rank, world, port, ret, exchange, banded, mode, C, N):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        n, q, K = 300, 2, 4
        row, col, val = _graph(n, 7, banded)
        rng = np.random.default_rng(8)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        g = rng.standard_normal((q, n, N)).astype(np.float32)
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device="cpu", exchange=exchange, ops=CpuOps())
        xl, gl = torch.from_numpy(x[:, sh.lo:sh.hi]).contiguous(), torch.from_numpy(g[:, sh.lo:sh.hi]).contiguous()
        gx, gW, gb2 = sh.layer_backward(xl, torch.from_numpy(W), gl, 2, mode)
        _, _, gb1 = sh.layer_backward(xl, torch.from_numpy(W), gl, 1, mode, needs=(False, False, True))
        T = sh.transpose()
        assert T.transpose() is sh and T.owned == sh.owned and T.lo == sh.lo
        L = O.coo_to_csr(row, col, val, n)
        rx, rW = O.layer_backward(L, x, W, g, "power" if mode == 0 else "chebyshev")
        ret[rank] = (_rel(gx.numpy(), rx[:, sh.lo:sh.hi]) * np.abs(rx[:, sh.lo:sh.hi]).max() / np.abs(rx).max(), _rel(gW.numpy(), rW),
                     _rel(gb1.numpy(), g.astype(np.float64).sum((0, 1))), _rel(gb2.numpy(), g.astype(np.float64).sum(0)[sh.lo:sh.hi]),
                     sh.use_project_first(C, N, K), T.exchange)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,banded,mode,C,N", [(2, "halo", True, 0, 20, 4), (3, "halo", True, 1, 20, 4), (2, "allgather", False, 1, 20, 4),
                                                            (2, "halo", True, 1, 5, 6), (3, "auto", False, 0, 5, 6), (3, "halo", True, 0, 4, 12)])
def test_sharded_gradients_match_the_oracle(world, exchange, banded, mode, C, N):
    """dX through the transposed shard (reverse halo exchange), dW all-reduced, both evaluation orders (2 N <= C: the adjoint terms on N-wide
    rows serve both gradients; otherwise recomputed basis + the layer on L^T, which itself projects first when 2 C <= N)"""
    res = _spawn(_grad_worker, world, exchange, banded, mode, C, N)
    for ex, eW, eb1, eb2, pf, _ in res:
        assert pf == (2 * N <= C)
        assert ex <= GRAD_TOL and eW <= GRAD_TOL and eb1 <= GRAD_TOL and eb2 <= GRAD_TOL, (ex, eW, eb1, eb2)


# ------------------------------------------------------------------------------------------------ module surface
def _module_worker(rank, world, port, ret, cls, banded):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        import tgcn_amd
        from tgcn_amd import dist as tdist
        n, q, K, g_ch = 240, 2, 4, 3
        row, col, val = _graph(n, 11, banded)
        Lsp = sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()           # duplicates summed: the operand every rank is handed
        rng = np.random.default_rng(12)
        # a "reference" state_dict: global shapes, as the single-GPU module (= the reference's class surface) holds them
        torch.manual_seed(5)
        if cls == "TGCNCheb_H":
            H, f = 14, 1
            single = tgcn_amd.TGCNCheb_H(Lsp, f, g_ch, K, H)
            x = rng.standard_normal((q, n, H)).astype(np.float32)
            torch.manual_seed(100 + rank)                                     # ranks start from DIFFERENT parameters: the first rank's are broadcast
            mod = tdist.ShardedTGCNCheb_H(Lsp, f, g_ch, K, H, exchange="auto", ops=CpuOps())
        elif cls == "TGCNCheb":
            f = 5
            single = tgcn_amd.TGCNCheb(Lsp, f, g_ch, K)
            x = rng.standard_normal((q, n, f)).astype(np.float32)
            torch.manual_seed(100 + rank)
            mod = tdist.ShardedTGCNCheb(Lsp, f, g_ch, K, exchange="halo", ops=CpuOps())
        else:
            f = 1
            single = tgcn_amd.GCNCheb(Lsp, f, g_ch, K)
            x = rng.standard_normal((q, n)).astype(np.float32)
            torch.manual_seed(100 + rank)
            mod = tdist.ShardedGCNCheb(Lsp, f, g_ch, K, exchange="allgather", ops=CpuOps())
        assert repr(mod).startswith("Sharded" + cls) and [k for k in mod.state_dict()] == [k for k in single.state_dict()]
        assert all(mod.state_dict()[k].shape == single.state_dict()[k].shape for k in single.state_dict())
        mod.load_state_dict(single.state_dict())                              # a reference state_dict loads as it is
        lo, hi = mod.owned_rows("cpu")
        xl = torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])).requires_grad_(True)
        out = mod(xl)
        gout = rng.standard_normal((q, n, g_ch)).astype(np.float32)
        out.backward(torch.from_numpy(np.ascontiguousarray(gout[:, lo:hi])))
        W = single.weight.detach().numpy()
        b = single.bias.detach().numpy()
        fwd = {"TGCNCheb_H": O.tgcn_cheb_h_forward, "TGCNCheb": O.tgcn_cheb_forward, "GCNCheb": O.gcn_cheb_forward}[cls]
        ref = fwd(Lsp.astype(np.float32), x, W, b)
        xr = x if cls != "GCNCheb" else x[:, :, None]
        rx, rW = O.layer_backward(Lsp.astype(np.float32), xr if cls != "TGCNCheb_H" else xr[..., None], W, gout, "power")
        rb = gout.astype(np.float64).sum(0)[None] if cls != "GCNCheb" else gout.astype(np.float64).sum((0, 1)).reshape(1, 1, -1)
        rxl = rx.reshape(x.shape)[:, lo:hi]
        ret[rank] = (float(np.abs(out.detach().numpy() - ref[:, lo:hi]).max() / np.abs(ref).max()),
                     float(np.abs(xl.grad.numpy() - rxl).max() / np.abs(rx).max()), _rel(mod.weight.grad.numpy(), rW), _rel(mod.bias.grad.numpy(), rb),
                     tuple(mod.bias.grad.shape), lo, hi, mod.shard("cpu").exchange, mod.shard("cpu").last_width)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cls,banded", [(2, "TGCNCheb_H", True), (3, "TGCNCheb", True), (2, "GCNCheb", False), (3, "TGCNCheb_H", False)])
def test_sharded_modules_keep_the_reference_surface(world, cls, banded):
    """constructor arguments of the reference + a group; weight / bias with their global shapes and names; forward(x_local) -> out_local;
    every gradient against the oracle; the full per-vertex bias gradient on every rank (all-reduced)"""
    res = _spawn(_module_worker, world, cls, banded)
    covered = np.zeros(240, np.int32)
    for e_out, e_x, e_W, e_b, bshape, lo, hi, _, width in res:
        assert e_out <= TOL and e_x <= GRAD_TOL and e_W <= GRAD_TOL and e_b <= GRAD_TOL, (e_out, e_x, e_W, e_b)
        assert bshape == ((1, 240, 3) if cls != "GCNCheb" else (1, 1, 3))
        covered[lo:hi] += 1
    assert np.all(covered == 1)
    if cls == "TGCNCheb_H":
        assert all(r[8] == 3 for r in res)           # H f = 14 floats in, 3 out: the hops and messages ran on 3-float rows (project-first)


# ------------------------------------------------------------------------------------------------ constructor: own rows only, tensors only
def _own_rows_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb, balanced_row_bounds
        n, q, C, N, K = 300, 2, 6, 5, 4
        row, col, val = _graph(n, 13, True)
        bounds = balanced_row_bounds(torch.as_tensor(row), n, world)          # agreed beforehand (any rank that sees the row counts can compute them)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        keep = (row >= lo) & (row < hi)                                       # this rank passes ONLY its own rows
        sh = VertexShardedCheb(n, torch.as_tensor(row[keep]), torch.as_tensor(col[keep]), torch.as_tensor(val[keep]), device="cpu", exchange="halo",
                               bounds=bounds, ops=CpuOps())
        rng = np.random.default_rng(14)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        out = sh.layer(torch.from_numpy(x[:, lo:hi]), torch.from_numpy(W), None, 0, 1)
        g = rng.standard_normal((q, n, N)).astype(np.float32)
        gx, gW, _ = sh.layer_backward(torch.from_numpy(x[:, lo:hi]).contiguous(), torch.from_numpy(W), torch.from_numpy(g[:, lo:hi]).contiguous(), 0, 1)
        L = O.coo_to_csr(row, col, val, n)
        ref = _ref_forward(L, x, W, None, 1)
        rx, rW = O.layer_backward(L, x, W, g, "chebyshev")
        ret[rank] = (float(np.abs(out.numpy() - ref[:, lo:hi]).max() / np.abs(ref).max()), float(np.abs(gx.numpy() - rx[:, lo:hi]).max() / np.abs(rx).max()),
                     _rel(gW.numpy(), rW), (sh.lo, sh.hi) == (lo, hi))
    finally:
        dist.destroy_process_group()


def test_a_rank_passes_only_its_own_rows():
    for e_out, e_x, e_W, same in _spawn(_own_rows_worker, 3):
        assert same and e_out <= TOL and e_x <= GRAD_TOL and e_W <= GRAD_TOL


def test_no_object_collective_in_the_shard_code():
    """index lists travel as tensors (count exchange + all_to_all_single), never as pickled Python objects"""
    src = open(os.path.join(os.path.dirname(__file__), "..", "tgcn_amd", "dist.py")).read()
    assert "all_gather_object" not in src and "broadcast_object" not in src and "gather_object" not in src


def test_row_bounds_can_respect_pool_groups():
    """gcn_pool_4 between two sharded layers (pytorch_hcp_tgcn.py:134-141) pools 4 consecutive vertices: row_multiple=4 keeps every group on one rank"""
    from tgcn_amd.dist import balanced_row_bounds
    g = torch.Generator().manual_seed(0)
    row = torch.randint(0, 1003, (9000,), generator=g)
    for world in (2, 3, 8):
        b = balanced_row_bounds(row, 1003, world, multiple=4).tolist()
        assert b[0] == 0 and b[-1] == 1003 and all(x % 4 == 0 for x in b[1:-1]) and all(b[i] <= b[i + 1] for i in range(world))
        plain = balanced_row_bounds(row, 1003, world).tolist()
        assert all(abs(x - y) <= 2 for x, y in zip(b, plain))


def _sync_init_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from tgcn_amd import dist as tdist
        n = 120
        row, col, val = _graph(n, 19, True)
        Lsp = sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()
        torch.manual_seed(1000 + rank)                        # every rank draws OTHER parameters ...
        mod = tdist.ShardedGCNCheb(Lsp, 2, 3, 3, ops=CpuOps(), row_multiple=4)
        before = mod.weight.detach().clone()
        sh = mod.shard("cpu")                                 # ... and the first rank's are broadcast when the shard is built
        ws = [torch.empty_like(mod.weight) for _ in range(world)]
        dist.all_gather(ws, mod.weight.detach().contiguous())
        ret[rank] = (all(torch.equal(w, ws[0]) for w in ws), bool(torch.equal(before, mod.weight.detach())), sh.lo % 4, sh.hi % 4 if sh.hi != n else 0)
    finally:
        dist.destroy_process_group()


def test_parameters_follow_the_groups_first_rank():
    res = _spawn(_sync_init_worker, 3)
    assert all(r[0] for r in res) and res[0][1] and not res[1][1] and not res[2][1]
    assert all(r[2] == 0 and r[3] == 0 for r in res)


def _model_worker(rank, world, port, ret):
    """a two-layer model written like the reference's (pytorch_hcp_tgcn.py:100-141: TGCNCheb_H -> relu -> GCNCheb -> relu -> head), its Chebyshev
    layers sharded: three SGD steps against the SAME model in one process on the whole graph (the single-GPU modules' arithmetic is replaced by
    float64 torch here -- this is the CPU suite -- so the comparison is sharded control flow + all-reduced gradients against plain autograd)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from tgcn_amd import dist as tdist
        n, q, H, g1, g2, K = 160, 3, 12, 5, 4, 3
        row, col, val = _graph(n, 23, True)
        Lsp = sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()
        torch.manual_seed(7)
        l1 = tdist.ShardedTGCNCheb_H(Lsp, 1, g1, K, H, ops=CpuOps(), exchange="halo")
        l2 = tdist.ShardedGCNCheb(Lsp, g1, g2, K, ops=CpuOps(), exchange="auto")
        params = list(l1.parameters()) + list(l2.parameters())
        # the same model on the whole graph, float64 dense autograd, from the same parameters
        Ld = torch.tensor(Lsp.toarray(), dtype=torch.float64)
        ref_p = [p.detach().double().clone().requires_grad_(True) for p in params]

        def cheb(x3, W, b):                    # dense-L classes' recursion (tgcn/nn/gcn.py:63-79)
            Xt, P = [x3], x3
            for k in range(1, W.shape[0]):
                P = torch.einsum("nm,qmc->qnc", Ld, P)
                Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
            return sum(Xt[k] @ W[k] for k in range(W.shape[0])) + b
        rng = np.random.default_rng(24)
        x = torch.as_tensor(rng.standard_normal((q, n, H)))
        y = torch.as_tensor(rng.standard_normal((q, n, g2)))
        lo, hi = l1.owned_rows("cpu")
        assert (lo, hi) == l2.owned_rows("cpu")
        losses = []
        for step in range(3):
            for p in params:
                p.grad = None
            h = torch.relu(l1(x[:, lo:hi].float()))
            out = l2(h)
            # mean over ALL vertices: every rank adds its rows' share; the gradient of the global loss w.r.t. the replicated parameters is the
            # all-reduced sum the layers produce
            loss_local = ((out - y[:, lo:hi].float()) ** 2).sum() / (q * n * g2)
            loss_local.backward()
            tot = loss_local.detach().clone()
            dist.all_reduce(tot)
            for rp in ref_p:
                rp.grad = None
            W1, b1, W2, b2 = ref_p
            hr = torch.relu(cheb(x, W1.reshape(K, H, g1), b1.reshape(1, n, g1)))
            lr = ((cheb(hr, W2, b2.reshape(1, 1, g2)) - y) ** 2).mean()
            lr.backward()
            errs = [_rel(p.grad.numpy(), rp.grad.numpy()) for p, rp in zip(params, ref_p)]
            losses.append((float(tot), float(lr), max(errs)))
            with torch.no_grad():
                for p, rp in zip(params, ref_p):
                    p -= 0.5 * p.grad
                    rp -= 0.5 * rp.grad
        ret[rank] = losses
    finally:
        dist.destroy_process_group()


def test_two_sharded_layers_train_like_the_whole_graph_model():
    res = _spawn(_model_worker, 3)
    for losses in res:
        for tot, ref, gerr in losses:
            assert abs(tot - ref) <= 1e-5 * abs(ref) and gerr <= 5e-5, losses
        assert losses[2][0] < losses[0][0]                      # and the steps do reduce the loss


# ------------------------------------------------------------------------------------------------ edge-list classes
def _edge_worker(rank, world, port, ret, timed, weighted):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tgcn_amd
        from tgcn_amd import dist as tdist
        rng = np.random.default_rng(31)
        n, E, q, f, g_ch, K, H = 220, 1500, 2, 3, 4, 4, 5
        ei = rng.integers(0, n, (2, E))
        ei[1, : E // 10] = ei[0, : E // 10]                      # self loops (removed, gcn.py:398)
        ei[:, E // 2: E // 2 + 100] = ei[:, :100]                # duplicate edges (separate terms)
        ei[0][ei[0] == 17] = 18                                  # vertex 17 has no outgoing edge: degree 0 -> deg^-1/2 = 0
        w = rng.uniform(0.5, 1.5, E).astype(np.float32) if weighted else None
        edge_index = torch.as_tensor(ei)
        edge_weight = None if w is None else torch.as_tensor(w)
        torch.manual_seed(3)
        if timed:
            single = tgcn_amd.ChebTimeConv(f, g_ch, K, H)
            x = rng.standard_normal((q, n, H, f)).astype(np.float32)
            torch.manual_seed(50 + rank)
            mod = tdist.ShardedChebTimeConv(f, g_ch, K, H, ops=CpuOps(), exchange="halo")
            fwd = O.cheb_time_conv_forward
        else:
            single = tgcn_amd.ChebConv(f, g_ch, K)
            x = rng.standard_normal((q, n, f)).astype(np.float32)
            torch.manual_seed(50 + rank)
            mod = tdist.ShardedChebConv(f, g_ch, K, ops=CpuOps(), exchange="auto")
            fwd = O.cheb_conv_forward
        mod.load_state_dict(single.state_dict())
        lo, hi = mod.owned_rows("cpu", edge_index, n, edge_weight)
        xl = torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])).requires_grad_(True)
        out = mod(xl, edge_index, edge_weight)                   # the cached shard of this edge_index: no num_vertices needed any more
        gout = rng.standard_normal((q, n, g_ch)).astype(np.float32)
        out.backward(torch.from_numpy(np.ascontiguousarray(gout[:, lo:hi])))
        W, b = single.weight.detach().numpy(), single.bias.detach().numpy()
        ref = fwd(x, ei, w, W, b)
        r_, c_, lap = O.edge_laplacian(ei, w, n)
        rx, rW = O.layer_backward(O.coo_to_csr(r_, c_, lap, n), x, W, gout, "chebyshev")
        ret[rank] = (float(np.abs(out.detach().numpy() - ref[:, lo:hi]).max() / np.abs(ref).max()),
                     float(np.abs(xl.grad.numpy() - rx[:, lo:hi]).max() / np.abs(rx).max()), _rel(mod.weight.grad.numpy(), rW),
                     _rel(mod.bias.grad.numpy(), gout.astype(np.float64).sum((0, 1))), lo, hi, len(mod._edge_shards))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,timed,weighted", [(2, False, False), (3, True, True), (2, True, False), (3, False, True)])
def test_sharded_edge_list_modules(world, timed, weighted):
    """ShardedChebConv / ShardedChebTimeConv (tgcn/nn/gcn.py:348-538; callers pygeo_hcp.py:85,129): global edge_index on every rank, self loops,
    duplicate edges, a degree-0 source, optional weights; output and all gradients against the oracle; one cached shard per edge_index"""
    res = _spawn(_edge_worker, world, timed, weighted)
    covered = np.zeros(220, np.int32)
    for e_out, e_x, e_W, e_b, lo, hi, nshards in res:
        assert e_out <= TOL and e_x <= GRAD_TOL and e_W <= GRAD_TOL and e_b <= GRAD_TOL, (e_out, e_x, e_W, e_b)
        assert nshards == 1
        covered[lo:hi] += 1
    assert np.all(covered == 1)


# ------------------------------------------------------------------------------------------------ random sweep
def _fuzz_worker(rank, world, port, ret, seed, cases):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        rng = np.random.default_rng(seed)                    # the same stream on every rank: the same cases
        out = []
        for case in range(cases):
            n = int(rng.integers(40, 260))
            q = int(rng.integers(1, 5))
            K = int(rng.choice([1, 2, 3, 5, 7]))
            C = int(rng.choice([1, 3, 4, 7, 16, 33]))
            N = int(rng.choice([1, 2, 5, 8, 17]))
            mode = int(rng.integers(0, 2))
            bias_kind = int(rng.integers(0, 3))
            exchange = str(rng.choice(["halo", "allgather", "auto"]))
            banded = bool(rng.integers(0, 2))
            depth = int(rng.integers(1, 4))
            row, col, val = _graph(n, int(rng.integers(0, 10 ** 6)), banded)
            if rng.integers(0, 2):                               # a block of vertices without entries (padded coarsened graphs, R-MAT)
                dead = rng.choice(n, n // 5, replace=False)
                keep = ~np.isin(row, dead)
                row, col, val = row[keep], col[keep], val[keep]
            x = rng.standard_normal((q, n, C)).astype(np.float32)
            W = (rng.standard_normal((K, C, N)) / 3).astype(np.float32)
            bias = None if bias_kind == 0 else (rng.standard_normal(N).astype(np.float32) if bias_kind == 1 else rng.standard_normal((n, N)).astype(np.float32))
            g = rng.standard_normal((q, n, N)).astype(np.float32)
            sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device="cpu", exchange=exchange, ops=CpuOps())
            xl = torch.from_numpy(np.ascontiguousarray(x[:, sh.lo:sh.hi]))
            bl = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias if bias_kind == 1 else bias[sh.lo:sh.hi]))
            Wt = torch.from_numpy(W)
            o1 = sh.layer(xl, Wt, bl, bias_kind, mode, depth=depth)
            same = bool(torch.equal(o1, sh.layer(xl, Wt, bl, bias_kind, mode, overlap=False)))
            o2 = sh.layer(xl, Wt, bl, bias_kind, mode, project_first=not sh.use_project_first(C, N, K))
            gx, gW, gb = sh.layer_backward(xl, Wt, torch.from_numpy(np.ascontiguousarray(g[:, sh.lo:sh.hi])), bias_kind, mode)
            L = O.coo_to_csr(row, col, val, n)
            ref = _ref_forward(L, x, W, bias, mode)
            rx, rW = O.layer_backward(L, x, W, g, "power" if mode == 0 else "chebyshev")
            s_o, s_x, s_w = max(np.abs(ref).max(), 1e-30), max(np.abs(rx).max(), 1e-30), max(np.abs(rW).max(), 1e-30)
            errs = [float(np.abs(o1.numpy() - ref[:, sh.lo:sh.hi]).max() / s_o), float(np.abs(o2.numpy() - ref[:, sh.lo:sh.hi]).max() / s_o),
                    float(np.abs(gx.numpy() - rx[:, sh.lo:sh.hi]).max() / s_x), float(np.abs(gW.numpy() - rW).max() / s_w)]
            if bias_kind:
                rb = g.astype(np.float64).sum((0, 1)) if bias_kind == 1 else g.astype(np.float64).sum(0)[sh.lo:sh.hi]
                errs.append(float(np.abs(gb.numpy() - rb).max() / max(np.abs(rb).max(), 1e-30)))
            out.append((dict(n=n, q=q, K=K, C=C, N=N, mode=mode, bias_kind=bias_kind, exchange=sh.exchange, banded=banded, depth=depth), same, errs))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,seed", [(2, 101), (3, 202)])
def test_random_sweep_of_the_sharded_layer(world, seed):
    """20 random shapes per world size: K = 1 ... 7, widths 1 ... 33 in and 1 ... 17 out (both evaluation orders each), both recurrences, three bias
    kinds, three exchange settings, pipeline depth 1 ... 3, graphs with and without a small cut and with vertices that have no entries -- forward in
    both orders, overlapped == plain bit for bit, all gradients against the oracle"""
    res = _spawn(_fuzz_worker, world, seed, 20)
    for per_rank in res:
        for cfg, same, errs in per_rank:
            assert same, cfg
            assert max(errs[:2]) <= TOL and max(errs[2:]) <= 5e-5, (cfg, errs)
