import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from tgcn_amd import functional as F, graph, _lib
from oracle import c_port
from test_compact_wave import _rmat_like, _dev
graph.COMPACT_MIN_ROWS = 1
q, C, N, K, bias_kind = 3, 64, 64, 5, 2
n = 40000
rng = np.random.default_rng(q * 100 + C + K)
row, col, val = _rmat_like(n, 50000, rng, True)
op = graph.GraphOperand.from_coo(n, _dev(row), _dev(col), _dev(val))
plan = op.compact_plan()
print("n_c", plan.n_c, "n_empty", plan.n_empty)
x = _dev(rng.standard_normal((q, n, C)).astype(np.float32))
W = _dev((rng.standard_normal((K, C, N)) / np.sqrt(K * C)).astype(np.float32))
bias = _dev(rng.standard_normal((n, N)).astype(np.float32))
Wt = F.fold_weight(F.power_fold_matrix(K, x.device), W)
W2 = Wt.reshape(K * C, N).contiguous()
e = op.edges.cpu().numpy()
ref = c_port.forward(0, op.rowptr.cpu().numpy(), np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1]).view(np.float32), x.cpu().numpy(), W.cpu().numpy(), bias.reshape(-1).cpu().numpy(), bias_kind)
rows, empty = plan.rows.cpu().numpy(), plan.empty.cpu().numpy()
def report(name, out):
    o = out.cpu().numpy()
    for b in range(q):
        d = np.abs(o[b] - ref[b])
        print("  %s sample %d: compact rows err %.2e (worst row %d)  empty rows err %.2e (worst row %d)" % (name, b, d[rows].max(), rows[d[rows].max(1).argmax()], d[empty].max(), empty[d[empty].max(1).argmax()]))
for pv in (4, 0, 1, 3):
    _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", pv))
    print("project_variant", pv)
    report("plain    ", F.cheb_forward_raw(op, x, W2, bias, bias_kind, F.MODE_POWER, K, layout=0, q_chunk=1))
    for qc in (1, 2, 3):
        report("compact%d " % qc, F.cheb_forward_compact(plan, x, W2, bias, bias_kind, K, q_chunk=qc))
