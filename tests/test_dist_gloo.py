"""world_size 2 / 3 gloo tests (CPU) of the N>1 paths.  The communication logic is the product code
(tgcn_amd/dist.py); the local compute is the oracle, injected through the hooks meant for exactly this."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cheb_oracle as O
from tools.cpu_standins import CpuOps       # numpy / scipy stand-ins for the HIP calls (test infrastructure)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(n, seed, banded):
    rng = np.random.default_rng(seed)
    if banded:      # small cut: neighbours within +-6 plus a few long-range edges
        row = np.repeat(np.arange(n), 5)
        col = np.clip(row + rng.integers(-6, 7, row.shape[0]), 0, n - 1)
        extra = rng.integers(0, n, (2, n // 20))
        row, col = np.concatenate([row, extra[0]]), np.concatenate([col, extra[1]])
    else:           # cut ~ everything
        row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
    row = np.concatenate([row, np.full(300, 7)])            # a hub row -> unbalanced row counts per shard
    col = np.concatenate([col, rng.integers(0, n, 300)])
    val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
    return row, col, val


def _worker(rank, world, port, exchange, banded, mode, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb, shard_time_steps
        n, q, C, N, K = 400, 3, 5, 4, 5
        row, col, val = _graph(n, 1, banded)
        rng = np.random.default_rng(2)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device="cpu",
                               exchange=exchange, ops=CpuOps())
        args = (torch.from_numpy(x[:, sh.lo:sh.hi]), torch.from_numpy(W), torch.from_numpy(bias[sh.lo:sh.hi]), 2, mode)
        out_local = sh.forward(*args)                              # overlapped: interior rows / other time steps under the exchange
        plain = sh.forward(*args, overlap=False)                   # one exchange, then one hop on all owned rows
        assert torch.equal(out_local, plain), "overlapped and plain forms differ"
        assert torch.equal(sh.forward(*args, depth=3), plain) and torch.equal(sh.forward(*args, depth=1), plain)
        if sh.exchange == "halo" and banded:
            assert 0 < sh.n_int < sh.owned                         # both classes of rows are exercised
        L = O.coo_to_csr(row, col, val, n)
        if mode == 1:
            basis = O.stack_chebyshev(L, x, K)
        else:
            P = [x]
            for _ in range(1, K):
                P.append(O._apply(L, P[-1]))
            basis = np.stack(P)
        ref = np.einsum("kqnc,kcg->qng", basis.astype(np.float64), W.astype(np.float64)) + bias
        err = np.abs(out_local.numpy() - ref[:, sh.lo:sh.hi]).max() / np.abs(ref).max()
        # every vertex is owned exactly once
        owned = torch.tensor([sh.owned])
        dist.all_reduce(owned)
        sl = shard_time_steps(7, rank, world)
        cnt = torch.tensor([sl.stop - sl.start])
        dist.all_reduce(cnt)
        ret[rank] = (float(err), sh.exchange, int(owned.item()), int(cnt.item()), sh.halo)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,banded,mode", [(2, "halo", True, 1), (3, "halo", True, 0), (2, "allgather", False, 1),
                                                        (3, "auto", False, 0), (2, "auto", True, 0)])
def test_vertex_sharded_matches_oracle(world, exchange, banded, mode):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), exchange, banded, mode, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        err, used, owned, cnt, halo = ret[rank]
        assert err <= 1e-5, (rank, err)
        assert owned == 400 and cnt == 7
        if exchange == "auto" and not banded:
            assert used == "allgather"


def _hybrid_worker(rank, world, port, exchange, banded, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb, hybrid_groups, shard_time_steps
        n, q, C, N, K = 300, 6, 4, 3, 4
        row, col, val = _graph(n, 5, banded)
        rng = np.random.default_rng(6)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        group, gi, ng = hybrid_groups(world, 2)
        sl = shard_time_steps(q, gi, ng)                       # this group's samples
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), group=group, device="cpu",
                               exchange=exchange, ops=CpuOps())
        out_local = sh.forward(torch.from_numpy(x[sl, sh.lo:sh.hi]), torch.from_numpy(W), torch.from_numpy(bias), 1, 1)
        assert torch.equal(out_local, sh.forward(torch.from_numpy(x[sl, sh.lo:sh.hi]), torch.from_numpy(W), torch.from_numpy(bias), 1, 1, overlap=False))
        L = O.coo_to_csr(row, col, val, n)
        ref = np.einsum("kqnc,kcg->qng", O.stack_chebyshev(L, x, K).astype(np.float64), W.astype(np.float64)) + bias
        err = np.abs(out_local.numpy() - ref[sl, sh.lo:sh.hi]).max() / np.abs(ref).max()
        ret[rank] = (float(err), gi, ng, sh.rank, sh.world, sl.start, sl.stop, sh.lo, sh.hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,banded", [("halo", True), ("allgather", False)])
def test_hybrid_vertex_x_time_layout(exchange, banded):
    """4 ranks = 2 time groups x 2 vertex shards: exchanges stay inside a group, groups never talk; together the ranks
    cover every (sample, vertex) exactly once."""
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hybrid_worker, args=(world, _free_port(), exchange, banded, ret), nprocs=world, join=True)
    assert len(ret) == world
    cover = np.zeros((6, 300), np.int32)
    for rank in range(world):
        err, gi, ng, vr, vw, s0, s1, lo, hi = ret[rank]
        assert err <= 1e-5, (rank, err)
        assert (gi, ng, vr, vw) == (rank // 2, 2, rank % 2, 2)
        cover[s0:s1, lo:hi] += 1
    assert np.all(cover == 1)
