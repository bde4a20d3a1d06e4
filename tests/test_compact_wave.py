"""Round-3 kernels of the large-graph path, through the C ABI, against the oracle:
  * wave segments (tgcn_csr_sched.seg_mode 1): the lane groups of a wave share one segment of a long row and fold their pieces
    inside the wave -- against lane-group segments (seg_mode 0) and the oracle, on rows around every length boundary;
  * compacted forward (tgcn_cheb_forward_compact_f32): hop tensors only for the vertices that have stored entries -- against
    the plain forward of the same operand (bitwise with one projection kernel) and against oracle/cheb_ref.c.
Tolerance: max|a-b| / max|b| <= 1e-5 per tensor, fp32."""
import numpy as np
import pytest
import torch

from conftest import golden_files, golden_ids, load_golden, rel_err
from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _graph_with_rows(n, lengths, rng, base_deg=5, empty=()):
    """random graph: `base_deg` entries per row, row i of `lengths` (dict row -> entries) overwritten, `empty` rows left empty"""
    deg = np.full(n, base_deg)
    for r, d in lengths.items():
        deg[r] = d
    deg[list(empty)] = 0
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.shape[0])
    val = (rng.standard_normal(row.shape[0]) / 4).astype(np.float32)
    return row, col, val


@pytest.mark.parametrize("C", [64, 16, 100, 8, 1])
def test_wave_segments_vs_oracle_and_group_segments(C, gpu_device, monkeypatch):
    from tgcn_amd import functional as F, graph, _lib
    n = 30000
    rng = np.random.default_rng(C)
    lanes = _lib.lib().tgcn_hop_lanes_per_row(C, 1)
    wave_len = 32 * (64 // lanes)
    lengths = {3: 33, 17: wave_len - 1, 18: wave_len, 19: wave_len + 1, 40: 2 * wave_len, 41: 3 * wave_len + 5, 900: 64 * wave_len + 7,
               901: 20000, 29999: 700, 12: 32, 13: 1}
    row, col, val = _graph_with_rows(n, lengths, rng, empty=(0, 7, 15000))
    L = O.coo_to_csr(row, col, val, n)
    x = rng.standard_normal((2, n, C)).astype(np.float32)
    z = rng.standard_normal((2, n, C)).astype(np.float32)
    ref = O._apply(L, x)
    outs = {}
    for mode in (0, 1):
        monkeypatch.setattr(graph, "SEG_MODE", mode)
        op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
        s = op.schedule_for(C, C % 4 == 0)
        assert s.seg_mode == (mode if s.lanes_per_row < 64 else 0)
        if mode == 1 and s.lanes_per_row < 64:
            assert s.seg_len == 32 * (64 // s.lanes_per_row) and s.nhuge >= 1 and s.nlong >= 4
        y, p = F.csr_hop(op, _dev(x), z=_dev(z), alpha=2.0, beta=-1.0, want_p=True)
        assert rel_err(p.cpu().numpy(), ref) <= TOL
        assert rel_err(y.cpu().numpy(), 2 * ref - z) <= TOL
        assert np.array_equal(y.cpu().numpy()[:, 7], -z[:, 7])
        assert torch.equal(y, F.csr_hop(op, _dev(x), z=_dev(z), alpha=2.0, beta=-1.0))      # run-to-run determinism
        # segment blocks in XCD-contiguous ranges: another place, the same sums
        _lib.check(_lib.lib().tgcn_set_tuning(b"hop_seg_remap", 1))
        try:
            assert torch.equal(y, F.csr_hop(op, _dev(x), z=_dev(z), alpha=2.0, beta=-1.0))
        finally:
            _lib.check(_lib.lib().tgcn_set_tuning(b"hop_seg_remap", 0))
        outs[mode] = p
    # rows that are ONE segment in both modes are summed in the same order
    short = np.array([r for r in range(n) if L.indptr[r + 1] - L.indptr[r] <= 32])
    assert torch.equal(outs[0][:, short], outs[1][:, short])


def _rmat_like(n, m, rng, symmetric=True):
    """skewed graph with many isolated vertices: endpoints drawn from a power law over a random relabelling"""
    perm = rng.permutation(n)
    u = perm[np.minimum((rng.random(m) ** 4 * n).astype(np.int64), n - 1)]
    v = perm[np.minimum((rng.random(m) ** 4 * n).astype(np.int64), n - 1)]
    keep = u != v
    u, v = u[keep], v[keep]
    if symmetric:
        row, col = np.concatenate([u, v]), np.concatenate([v, u])
    else:
        row, col = u, v
    val = (rng.standard_normal(row.shape[0]) / 6).astype(np.float32)
    return row, col, val


@pytest.mark.parametrize("symmetric", [True, False], ids=["symmetric", "entries-into-empty-rows"])
@pytest.mark.parametrize("q,C,N,K,bias_kind", [(3, 64, 64, 5, 2), (2, 32, 48, 3, 1), (1, 20, 8, 2, 0), (2, 64, 64, 6, 2)])
def test_compact_forward_equals_plain_forward(q, C, N, K, bias_kind, symmetric, gpu_device, monkeypatch):
    from tgcn_amd import functional as F, graph, _lib
    from oracle import c_port
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    n = 40000
    rng = np.random.default_rng(q * 100 + C + K)
    row, col, val = _rmat_like(n, 50000, rng, symmetric)
    op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    plan = op.compact_plan()
    assert plan is not None and plan.n_c + plan.n_empty == n and plan.n_empty > n // 8
    deg = np.bincount(row, minlength=n)
    assert np.array_equal(plan.rows.cpu().numpy(), np.flatnonzero(deg > 0)) and np.array_equal(plan.empty.cpu().numpy(), np.flatnonzero(deg == 0))
    if not symmetric:      # some entries point at vertices without outgoing entries: they must gather the zero row
        assert (deg[col] == 0).any()
    x = _dev(rng.standard_normal((q, n, C)).astype(np.float32))
    W = _dev((rng.standard_normal((K, C, N)) / np.sqrt(K * C)).astype(np.float32))
    bias = None if bias_kind == 0 else _dev(rng.standard_normal((N,) if bias_kind == 1 else (n, N)).astype(np.float32))
    Wt = F.fold_weight(F.power_fold_matrix(K, x.device), W) if K > 2 else W
    W2 = Wt.reshape(K * C, N).contiguous()
    _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", 4))      # one (exact fp32) projection kernel for both: bitwise comparable
    try:
        plain = F.cheb_forward_raw(op, x, W2, bias, bias_kind, F.MODE_POWER, K, layout=0, q_chunk=1)
        comp = F.cheb_forward_compact(plan, x, W2, bias, bias_kind, K, q_chunk=1)
        comp2 = F.cheb_forward_compact(plan, x, W2, bias, bias_kind, K, q_chunk=2)
        _lib.check(_lib.lib().tgcn_set_tuning(b"compact_proj", 1))      # ONE projection over all vertices, hop tensors through the id map
        comp3 = F.cheb_forward_compact(plan, x, W2, bias, bias_kind, K, q_chunk=2)
    finally:
        _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", 0))
        _lib.check(_lib.lib().tgcn_set_tuning(b"compact_proj", 0))
    assert torch.equal(plain, comp) and torch.equal(comp, comp2) and torch.equal(comp, comp3)
    assert np.array_equal(plan.cid.cpu().numpy()[plan.rows.cpu().numpy()], np.arange(plan.n_c)) and (plan.cid[plan.empty.long()] == plan.n_c).all()
    # default kernels, through the dispatcher, against the C restatement of the reference's algorithm (unfolded weights)
    out = F.layer_forward(op, x, W, F.power_fold_matrix(K, x.device) if K > 2 else None, bias, bias_kind, F.MODE_POWER)
    rowptr = op.rowptr.cpu().numpy()
    e = op.edges.cpu().numpy()
    b = np.zeros(1, np.float32) if bias is None else bias.reshape(-1).cpu().numpy()
    ref = c_port.forward(0, rowptr, np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1]).view(np.float32), x.cpu().numpy(),
                         W.cpu().numpy(), b, bias_kind)
    assert rel_err(out.cpu().numpy(), ref) <= TOL


def test_compact_plan_is_declined_without_empty_rows(gpu_device, monkeypatch):
    from tgcn_amd import graph
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    n = 5000
    row = np.repeat(np.arange(n), 3)
    col = (row + np.tile([1, 2, 3], n)) % n
    op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(np.ones(row.shape[0], np.float32)))
    assert op.compact_plan() is None


_PAD = [p for p in golden_files("") if "pad48" in p and ("GCNCheb_" in p or "TGCNCheb" in p)]


@pytest.mark.parametrize("path", _PAD, ids=golden_ids(_PAD))
def test_compact_forward_on_padded_fixtures(path, gpu_device, monkeypatch):
    """the reference's own case of isolated vertices: coarsening pads graphs with fake vertices (gcn/coarsening.py:167-217);
    fixtures *_pad48_* carry 48 of them.  Forced through the compacted general path, against the reference's output."""
    from tgcn_amd import functional as F, graph
    from test_hip_parity import _make_layer
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    monkeypatch.setattr(graph, "COMPACT_MIN_EMPTY", 0.0)
    monkeypatch.setattr(F, "SMALL_PATH", False)
    monkeypatch.setattr(F, "PROJECT_FIRST", False)
    monkeypatch.setattr(F, "choose_layout", lambda q, n, C_row: 0)      # the compacted path works on the (q, n, C) layout
    g = load_golden(path)
    layer, extra = _make_layer(g)
    x = _dev(g["x"])
    used = []
    real = F.cheb_forward_compact
    monkeypatch.setattr(F, "cheb_forward_compact", lambda *a, **k: (used.append(1), real(*a, **k))[1])
    with torch.no_grad():        # inference: a training forward keeps the full-label hop tensors for its backward instead
        out = layer(x, *extra)
    assert rel_err(out.detach().cpu().numpy(), g["out"]) <= TOL
    if layer.weight.shape[0] >= 2:
        assert used, "the compacted path was not taken"


def test_offsets_beyond_2_31_elements(gpu_device):
    """batch * n * C > 2^31 elements through the layer driver with q_chunk = 1 (the shape of the headline run: sample t of cfg5 sits
    t * 640 M floats into x and out).  Size-independent properties, all compared on the device:
      * the last sample computed inside the batch == the same sample computed alone at offset 0 (bitwise: same kernels, samples
        are independent), for the plain and for the compacted forward;
      * plain and compacted forward agree; linearity of the layer without bias on the last sample."""
    from tgcn_amd import functional as F, graph, _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    q, n, C, N, K = 9, 4_000_000, 64, 64, 3
    assert q * n * C > 2 ** 31
    m = 20_000_000
    live = n // 2                                                  # the other half of the vertices keeps no entry
    u = (torch.rand(m, device="cuda", generator=g) ** 3 * live).long().clamp_(max=live - 1)
    v = (torch.rand(m, device="cuda", generator=g) ** 3 * live).long().clamp_(max=live - 1)
    perm = torch.randperm(n, device="cuda", generator=g)
    row, col = torch.cat([perm[u], perm[v]]), torch.cat([perm[v], perm[u]])
    val = torch.randn(row.numel(), device="cuda", generator=g) * 0.05
    op = graph.GraphOperand.from_coo(n, row, col, val)
    del row, col, val, u, v
    plan = op.compact_plan()
    assert plan is not None and plan.n_empty > n // 8
    x = torch.randn(q, n, C, device="cuda", generator=g)
    W = torch.randn(K * C, N, device="cuda", generator=g) / (K * C) ** 0.5
    bias = torch.randn(n, N, device="cuda", generator=g)
    for fwd in (lambda xx, b, k: F.cheb_forward_raw(op, xx, W, b, k, F.MODE_POWER, K, layout=0, q_chunk=1),
                lambda xx, b, k: F.cheb_forward_compact(plan, xx, W, b, k, K, q_chunk=1)):
        full = fwd(x, bias, 2)
        for t in (0, q - 1):
            alone = fwd(x[t:t + 1].clone(), bias, 2)
            assert torch.equal(full[t], alone[0]), "sample %d differs when computed inside the batch" % t
        last = full[q - 1].clone()
        del full, alone
        if "ref_last" not in locals():
            ref_last = last
        else:
            assert float((last - ref_last).abs().max() / ref_last.abs().max()) <= 1e-6
    # the hop itself with the batch inside ONE launch (in-kernel b * batch_stride beyond 2^31 elements): last sample == computed alone,
    # and the adjoint identity <L x, y> = <x, L^T y> on that sample
    yb = F.csr_hop(op, x)
    y8 = F.csr_hop(op, x[q - 1:q].clone())
    assert torch.equal(yb[q - 1], y8[0])
    w = torch.randn(1, n, C, device="cuda", generator=g)
    lhs_a = (y8.double() * w.double()).sum()
    rhs_a = (x[q - 1:q].double() * F.csr_hop(op.transpose(), w).double()).sum()
    assert abs(lhs_a - rhs_a) <= 1e-6 * max(abs(lhs_a), abs(rhs_a), 1.0)
    del yb, y8, w
    # linearity on the last sample: A(0.5 a + b) = 0.5 A a + A b
    a, b = x[q - 1:q], x[0:1]
    lhs = F.cheb_forward_compact(plan, 0.5 * a + b, W, None, 0, K)
    rhs = 0.5 * F.cheb_forward_compact(plan, a.clone(), W, None, 0, K) + F.cheb_forward_compact(plan, b.clone(), W, None, 0, K)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= TOL


@pytest.mark.parametrize("M", [70000, 90000, 300])
def test_wide_bf16x3_projection_tail_tiles(M, gpu_device):
    """project_x3v2_kernel hands the rows of a thinly filled last round out as 128-row tiles (cfg4: 352 tiles = 256 + 96 on 256
    CUs): same arithmetic per row, so bitwise the result of 256-row tiles throughout, and the fp64 product within tolerance."""
    from tgcn_amd import functional as F, _lib
    rng = np.random.default_rng(M)
    Kc, N, T = 72, 160, 2
    terms = [rng.standard_normal((M, Kc)).astype(np.float32) for _ in range(T)]
    W = (rng.standard_normal((T, Kc, N)) / np.sqrt(T * Kc)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = sum(t.astype(np.float64) @ w.astype(np.float64) for t, w in zip(terms, W)) + bias
    L = _lib.lib()
    _lib.check(L.tgcn_set_tuning(b"project_variant", 3))
    try:
        outs = []
        for tail in (1, 0):
            _lib.check(L.tgcn_set_tuning(b"x3_tail", tail))
            outs.append(F.cheb_project([_dev(t) for t in terms], _dev(W), _dev(bias), 1, M))
    finally:
        _lib.check(L.tgcn_set_tuning(b"x3_tail", 1))
        _lib.check(L.tgcn_set_tuning(b"project_variant", 0))
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0].cpu().numpy(), ref) <= TOL


def test_hop_streaming_form_keeps_the_result(gpu_device):
    """Outputs larger than the Infinity Cache take hop_kernel's form with non-temporal entry loads, row stores and partial-row
    stores (tgcn_set_tuning("hop_stream")): cache hints only -- bitwise the result of the plain form, long rows and fix-up included."""
    from tgcn_amd import functional as F, graph, _lib
    g = torch.Generator(device="cuda").manual_seed(11)
    n, m, C = 1_200_000, 8_000_000, 64
    assert n * C * 4 > 256 << 20
    row = torch.randint(0, n, (m,), device="cuda", generator=g)
    col = (torch.rand(m, device="cuda", generator=g) ** 3 * n).long().clamp_(max=n - 1)
    row[: 200_000] = torch.randint(0, 40, (200_000,), device="cuda", generator=g)          # long rows: segments + partial rows + fix-up
    val = torch.randn(m, device="cuda", generator=g) * 0.1
    op = graph.GraphOperand.from_coo(n, row, col, val)
    s = op.schedule_for(C)
    assert s.nlong > 0 and s.npartial > 0
    x = torch.randn(1, n, C, device="cuda", generator=g)
    z = torch.randn(1, n, C, device="cuda", generator=g)
    L = _lib.lib()
    outs = []
    for on in (1, 0):
        _lib.check(L.tgcn_set_tuning(b"hop_stream", on))
        try:
            outs.append(F.csr_hop(op, x, z=z, alpha=2.0, beta=-1.0, want_p=True))
        finally:
            _lib.check(L.tgcn_set_tuning(b"hop_stream", 1))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and the plain form is the oracle-checked one: spot-check 2000 rows against a float64 row-by-row sum
    rows = torch.randint(0, n, (2000,), device="cuda", generator=g).tolist() + list(range(40))
    rp, e = op.rowptr.cpu().numpy(), op.edges.cpu().numpy()
    xc = x[0].cpu().numpy().astype(np.float64)
    got = outs[0][1][0].cpu().numpy()
    for r in rows[:300] + rows[-40:]:
        cols, vals = e[rp[r]: rp[r + 1], 0], e[rp[r]: rp[r + 1], 1].copy().view(np.float32).astype(np.float64)
        ref = (vals[:, None] * xc[cols]).sum(0)
        assert np.abs(got[r] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-3)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 4: compact hop tensors for the Chebyshev recurrence (ChebConv / ChebTimeConv) and for the training forward + backward
# ---------------------------------------------------------------------------------------------------------------------------------
_MODULE_FILES = [p for pre in ("GCNCheb_", "TGCNCheb_", "TGCNChebH_", "ChebConv_", "ChebTimeConv_") for p in golden_files(pre)]


@pytest.mark.parametrize("path", _MODULE_FILES, ids=golden_ids(_MODULE_FILES))
def test_compact_training_path_on_every_module_fixture(path, gpu_device, monkeypatch):
    """Every module fixture of the reference (all five classes; self loops, degree-0 sources, isolated padded vertices, edge weights,
    K = 1 ... 25) FORCED through the compact path -- forward kept for training, weight gradient from the compact terms, input gradient
    as the compact layer on L^T -- against the reference's own outputs AND its own autograd gradients."""
    from tgcn_amd import functional as F, graph
    from test_hip_parity import _make_layer, GRAD_TOL
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    monkeypatch.setattr(graph, "COMPACT_MIN_EMPTY", 0.0)
    monkeypatch.setattr(F, "SMALL_PATH", False)
    monkeypatch.setattr(F, "PROJECT_FIRST", False)
    monkeypatch.setattr(F, "choose_layout", lambda q, n, C_row: 0)
    monkeypatch.setattr(F, "_pad_rows", lambda op, x3, w, mode: (x3, w))      # odd row lengths too (scalar-load forms)
    g = load_golden(path)
    layer, extra = _make_layer(g)
    used = {"fwd": 0, "wgrad": 0, "drv": 0}
    for name, key in (("compact_forward", "fwd"), ("compact_wgrad", "wgrad"), ("cheb_forward_compact", "drv")):
        real = getattr(F, name)
        monkeypatch.setattr(F, name, (lambda real, key: lambda *a, **k: (used.__setitem__(key, used[key] + 1), real(*a, **k))[1])(real, key))
    x = _dev(g["x"]).requires_grad_(True)
    out = layer(x, *extra)
    assert rel_err(out.detach().cpu().numpy(), g["out"]) <= TOL
    out.backward(_dev(g["grad_out"]))
    assert rel_err(x.grad.cpu().numpy(), g["grad_x"]) <= GRAD_TOL
    assert rel_err(layer.weight.grad.cpu().numpy(), g["grad_weight"]) <= GRAD_TOL
    if int(g["has_bias"]):
        assert rel_err(layer.bias.grad.cpu().numpy(), g["grad_bias"]) <= GRAD_TOL
    K = layer.weight.shape[0]
    if 2 <= K <= 32:
        assert used["fwd"] >= 1 and used["wgrad"] == 1 and used["fwd"] + used["drv"] >= 2, used     # forward, dW, and dx on L^T
    # inference of the same module (no basis kept) agrees with the training forward
    with torch.no_grad():
        assert rel_err(layer(_dev(g["x"]), *extra).cpu().numpy(), out.detach().cpu().numpy()) <= TOL


def _directed_with_isolated(n, m, rng):
    """skewed DIRECTED edge list over a random relabelling: vertices with entries, vertices that are only pointed at, isolated ones"""
    perm = rng.permutation(n)
    live = n // 2
    u = perm[np.minimum((rng.random(m) ** 3 * (live // 2)).astype(np.int64), live // 2 - 1)]              # sources: a quarter of the vertices
    v = perm[np.minimum((rng.random(m) ** 3 * live).astype(np.int64), live - 1)]                           # targets: half of them
    keep = u != v
    return u[keep], v[keep]


@pytest.mark.parametrize("symmetric", [True, False], ids=["symmetric", "referenced-only-vertices"])
@pytest.mark.parametrize("cls,q,f,g_out,K", [("ChebConv", 2, 16, 24, 5), ("ChebConv", 1, 64, 64, 3), ("ChebTimeConv", 2, 2, 8, 4), ("GCNCheb", 2, 32, 16, 5),
                                             ("GCNCheb", 16, 1, 64, 5), ("ChebConv", 5, 4, 8, 3)],      # the last two: vertex-major layout 1 (short per-sample rows, cfg5n's shape)
                         ids=["ChebConv-q2-f16", "ChebConv-q1-f64", "ChebTimeConv-q2-f2", "GCNCheb-q2-f32", "GCNCheb-q16-f1-layout1", "ChebConv-q5-f4-layout1"])
def test_compact_layers_equal_uncompacted_layers(cls, q, f, g_out, K, symmetric, gpu_device, monkeypatch):
    """Both recursions on a 70 k-vertex graph with isolated vertices (and, unsymmetric, vertices that entries point at but that have
    none of their own: for the Chebyshev recurrence those stay in the compact set, T_k of such a vertex is +-x, not 0): the compact
    forward / backward against the same module with compaction switched off, and the forward against the oracle."""
    import tgcn_amd
    from tgcn_amd import functional as F
    monkeypatch.setattr(F, "COMPACT_LAYOUT1", True)       # the vertex-major form is built and tested, off by default (slower on cfg5n)
    rng = np.random.default_rng(K * 7 + f)
    n = 70_000
    u, v = _directed_with_isolated(n, 300_000, rng)
    if symmetric:
        u, v = np.concatenate([u, v]), np.concatenate([v, u])
    H = 6
    torch.manual_seed(3)
    if cls == "GCNCheb":
        deg = np.bincount(u, minlength=n).astype(np.float64)
        dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
        val = (-dis[u] * dis[v]).astype(np.float32)
        op = tgcn_amd.GraphOperand.from_coo(n, _dev(u), _dev(v), _dev(val))
        layer, extra = tgcn_amd.GCNCheb(op, f, g_out, K).cuda(), ()
        x = rng.standard_normal((q, n, f)).astype(np.float32)
        mode, plan = F.MODE_POWER, op.compact_plan("rows")
    else:
        ei = _dev(np.stack([u, v]))
        extra = (ei,)
        if cls == "ChebConv":
            layer = tgcn_amd.ChebConv(f, g_out, K).cuda()
            x = rng.standard_normal((q, n, f)).astype(np.float32)
        else:
            layer = tgcn_amd.ChebTimeConv(f, g_out, K, H).cuda()
            x = rng.standard_normal((q, n, H, f)).astype(np.float32)
        op = layer._operand(_dev(x), ei, None)
        mode, plan = F.MODE_CHEBYSHEV, op.compact_plan("closed")
    assert plan is not None and plan.n_empty >= n // 8
    assert F.choose_layout(q, n, x.reshape(q, n, -1).shape[2]) == (1 if (q > 1 and x.reshape(q, n, -1).shape[2] < 32) else 0)
    if mode == F.MODE_CHEBYSHEV and not symmetric:
        assert plan.n_c > op.compact_plan("rows").n_c          # referenced-only vertices are kept for the Chebyshev recurrence
    gout = rng.standard_normal((q, n, g_out)).astype(np.float32)
    res = {}
    for compact in (True, False):
        monkeypatch.setattr(F, "COMPACT", compact)
        layer.zero_grad()
        xd = _dev(x).requires_grad_(True)
        out = layer(xd, *extra)
        out.backward(_dev(gout))
        res[compact] = [t.detach().cpu().numpy() for t in (out, xd.grad, layer.weight.grad, layer.bias.grad)]
    for a, b in zip(res[True], res[False]):
        assert rel_err(a, b) <= 2e-5
    # forward against the oracle
    Ls = op.to_scipy()
    W = layer.weight.detach().cpu().numpy().reshape(K, -1, g_out)
    x3 = x.reshape(q, n, -1)
    if mode == F.MODE_POWER:
        ref = O.gcn_cheb_forward(Ls, x3, W, layer.bias.detach().cpu().numpy())
    else:
        T = [x3, O._apply(Ls, x3)]
        for k in range(2, K):
            T.append(2 * O._apply(Ls, T[k - 1]) - T[k - 2])
        ref = sum(T[k] @ W[k] for k in range(K)) + layer.bias.detach().cpu().numpy()
    assert rel_err(res[True][0], ref) <= TOL


@pytest.mark.parametrize("symmetric", [True, False], ids=["symmetric", "entries-into-empty-rows"])
@pytest.mark.parametrize("q,C,N,K,bias_kind,q_chunk", [(3, 64, 64, 5, 2, 1), (4, 64, 64, 3, 2, 2), (2, 64, 32, 2, 1, 1), (1, 32, 16, 4, 0, 1), (2, 128, 64, 3, 2, 2)])
def test_last_hop_fused_into_the_projection_is_bitwise_the_unfused_forward(q, C, N, K, bias_kind, q_chunk, symmetric, gpu_device, monkeypatch):
    """the compacted driver gathers the rows of at most 32 entries of the LAST hop inside the projection
    (project_x3_gather_kernel) and runs the hop launch for the longer rows only; the last hop tensor is not written for the others.
    Same per-row arithmetic (stored-order fmaf chain, same bf16x3 split and MFMA order) => bitwise the result of hop + projection
    (tgcn_set_tuning("fuse_last_hop", 0)), and within 1e-5 of the oracle.  K = 2: the fused hop gathers from x through the caller-label operand."""
    from tgcn_amd import functional as F, graph, _lib
    monkeypatch.setattr(graph, "COMPACT_MIN_ROWS", 1)
    n = 40000
    rng = np.random.default_rng(q * 10 + C + K)
    row, col, val = _rmat_like(n, 60000, rng, symmetric)
    op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
    plan = op.compact_plan()
    assert plan is not None
    sched = plan.schedule_for(C, True)
    assert sched.nseg > 0 and sched.nlong > 0           # rows above the threshold exist: they take the hop launch, the others the gather
    x = _dev(rng.standard_normal((q, n, C)).astype(np.float32))
    W = _dev((rng.standard_normal((K, C, N)) / np.sqrt(K * C)).astype(np.float32))
    bias = None if bias_kind == 0 else _dev(rng.standard_normal((N,) if bias_kind == 1 else (n, N)).astype(np.float32))
    W2 = W.reshape(K * C, N).contiguous()
    L = _lib.lib()
    _lib.check(L.tgcn_set_tuning(b"project_variant", 3))          # the bf16x3 kernel at this size too (auto takes it from 8192 rows)
    try:
        outs = []
        for fuse in (1, 0):
            _lib.check(L.tgcn_set_tuning(b"fuse_last_hop", fuse))
            _lib.profile_start(256)
            outs.append(F.cheb_forward_compact(plan, x, W2, bias, bias_kind, K, q_chunk=q_chunk))
            prof = _lib.profile_stop(256)
            full, long_only, gathered = (sum(1 for kind, _ in prof if kind == k) for k in (0, 7, 8))
            # fused: one hop launch per time step covers the rows above the threshold only, and a projection per pass gathers the rest
            assert (full, long_only) == ((q * (K - 2), q) if fuse else (q * (K - 1), 0)), (fuse, full, long_only)
            assert gathered == ((q + q_chunk - 1) // q_chunk if fuse else 0)
    finally:
        L.tgcn_reset_tuning()
    assert torch.equal(outs[0], outs[1])
    Ls = op.to_scipy()
    P = [x.cpu().numpy()]
    for _ in range(1, K):
        P.append(O._apply(Ls, P[-1]))
    ref = sum(P[k].astype(np.float64) @ W[k].cpu().numpy().astype(np.float64) for k in range(K))
    if bias is not None:
        ref = ref + bias.cpu().numpy()
    assert rel_err(outs[0].cpu().numpy(), ref) <= TOL
