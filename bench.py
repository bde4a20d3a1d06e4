#!/usr/bin/env python3
"""bench.py -- Cheb-TGCN forward on MI355X: G edge.timesteps/s + achieved HBM GB/s against the roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg5|cfg4|cfg3|cfg2] [--labeling random|degree|natural]

A "step" is ONE forward of the layer over one batch of synthetic input resident in HBM:
  cfg5 (default, BASELINE.json's metric): R-MAT n=10M / nnz=160M, TGCNCheb(L, 64, 64, K=5) semantics with q=T=16
        time steps  -> 4 hops x 16 time steps of CSR x (n x 64) + the (K*64) x 64 projection.
  cfg4: sheet mesh n=90k / nnz~0.9M, TGCNCheb_H(L, 1, 32, 5, 1200), q=1.
  cfg3: MNIST grid n=784, TGCNCheb_H(L, 1, 64, 5, 28), q=64.      cfg2: GCNCheb(L, 1, 64, 5), q=128.
N > 1 (driver-launched with torch.distributed.run): STRONG scaling on the named shape -- the q = T = 16 time steps of the
workload are split over the ranks (16/N each; every rank holds the CSR, no data-path collective), value is the whole-job
aggregate.  --shard vertex / hybrid run the vertex-sharded layer (halo / all-gather exchange per hop) on the same shape;
--scaling weak keeps 16 time steps per rank instead.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = hop_kernel,
algorithmic bytes per launch / mean launch duration from hipEvents recorded around every hop launch of the timed
steps) and `cpu_baseline` (oracle/cheb_ref.c, OpenMP, timed on this box's host cores on a bounded sample).
`roofline.physical_*`: the same for the bytes a launch physically has to move once (entries, row pointers, and only the rows that exist in
the hop tensors -- compaction is fewer rows processed, not more bandwidth); `roofline.mean_launch_ms_by_hop`: by hop position.
N > 1: the vertex-sharded / hybrid extras (`other_shardings`) size themselves to --extras-budget and never decide the exit code; they run through
the sharded MODULES (tgcn_amd.dist.ShardedTGCNCheb / ShardedTGCNCheb_H).  `--shard vertex` without a launcher forms a one-rank group itself.
The default run (cfg5, one GPU) also carries `other_workloads` -- the R-MAT with degree-sorted labels, cfg2 (f = 1 and 64), cfg3, cfg4, each with ms
per forward, its dominant kernel's roofline and the GPU-vs-CPU errors, measured after the headline with its memory freed (--no-others skips them) --
and times the CPU baseline on as many time steps as --cpu-budget seconds allow (`cpu_baseline.samples_timed`, `scaled_from`, `host`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

def baseline_metric():
    """The metric string of BASELINE.json, verbatim (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "Cheb-TGCN fwd: G edge\u00b7timesteps/s + achieved HBM GB/s, K=5 on 160M-edge graph"


HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)


def build_workload(name, labeling, device, n_override=None, nnz_override=None):
    from tools import synth
    from tgcn_amd.graph import GraphOperand
    if name == "cfg5":
        n, nnz = n_override or 10_000_000, nnz_override or 160_000_000
        _, row, col, val = synth.rmat(n, nnz, seed=12345, labeling=labeling, device=device)
        spec = dict(cls="TGCNCheb", q=16, H=1, f=64, g=64, K=5,
                    desc="R-MAT(0.57,0.19,0.19,0.05) n=%d nnz=%d %s labels, TGCNCheb(L,64,64,K=5), q=T=16" % (n, nnz, labeling))
    elif name == "cfg5n":
        # SURVEY.md 8d "narrow variant": same graph, one input channel per time step (F = 16 floats per vertex and hop)
        n, nnz = n_override or 10_000_000, nnz_override or 160_000_000
        _, row, col, val = synth.rmat(n, nnz, seed=12345, labeling=labeling, device=device)
        spec = dict(cls="TGCNCheb", q=16, H=1, f=1, g=64, K=5,
                    desc="R-MAT(0.57,0.19,0.19,0.05) n=%d nnz=%d %s labels, TGCNCheb(L,1,64,K=5), q=T=16 (narrow)" % (n, nnz, labeling))
    elif name == "cfg4":
        n, row, col, val = synth.sheet_mesh(300, device=device)
        spec = dict(cls="TGCNCheb_H", q=1, H=1200, f=1, g=32, K=5, desc="sheet mesh n=90000 nnz=%d, TGCNCheb_H(L,1,32,5,1200), q=1" % row.numel())
    elif name == "hcp148":
        # the reference's own HCP model shape: TGCNCheb_H(L, 1, 32, 10, 15) on the 148-parcel DTI graph, batch 512
        # (examples/pytorch_based/pytorch_hcp_tgcn.py:103-104,242); graph = fixture generated from load/res/*.mat
        z = np.load(os.path.join(ROOT, "tests", "golden", "TGCNChebH_dti148_q4_f1_g32_K10_H15.npz"))
        n = int(z["n"])
        rowptr = torch.as_tensor(z["rowptr"])
        row = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1]).to(device)
        col, val = torch.as_tensor(z["col"]).long().to(device), torch.as_tensor(z["val"]).to(device)
        spec = dict(cls="TGCNCheb_H", q=512, H=15, f=1, g=32, K=10, desc="HCP aparc DTI graph n=148 nnz=%d (dense), TGCNCheb_H(L,1,32,10,15), q=512" % col.numel())
    elif name in ("cfg3", "cfg2", "cfg2w"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "GCNCheb_grid784_q3_f1_g8_K5_x2d.npz"))
        n = int(z["n"])
        rowptr = torch.as_tensor(z["rowptr"])
        row = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1]).to(device)
        col, val = torch.as_tensor(z["col"]).long().to(device), torch.as_tensor(z["val"]).to(device)
        if name == "cfg3":
            spec = dict(cls="TGCNCheb_H", q=64, H=28, f=1, g=64, K=5, desc="MNIST 8-NN grid n=784 nnz=6396, TGCNCheb_H(L,1,64,5,28), q=64")
        elif name == "cfg2w":   # the f = 64 variant of cfg2 (a second layer)
            spec = dict(cls="GCNCheb", q=128, H=1, f=64, g=64, K=5, desc="MNIST 8-NN grid n=784 nnz=6396, GCNCheb(L,64,64,5), q=128")
        else:
            spec = dict(cls="GCNCheb", q=128, H=1, f=1, g=64, K=5, desc="MNIST 8-NN grid n=784 nnz=6396, GCNCheb(L,1,64,5), q=128")
    else:
        raise SystemExit("unknown workload " + name)
    # one-off operand construction, timed for the record (not part of a step): COO -> CSR, then the hop schedule of the layer's row width
    from tgcn_amd import graph as _graph
    sync = (lambda: torch.cuda.synchronize()) if torch.device(device).type == "cuda" else (lambda: None)
    if torch.device(device).type == "cuda" and n * 8 < (1 << 31):
        # untimed first build: the caching allocator obtains its blocks from the driver (a cold hipMalloc of gigabytes takes ~0.1 s)
        op = GraphOperand.from_coo(n, row, col, val, device)
        del op
    sync()
    t0 = time.perf_counter()
    op = GraphOperand.from_coo(n, row, col, val, device)
    sync()
    t1 = time.perf_counter()
    del row, col, val
    spec["operand_build"] = dict(builder=_graph.BUILDER if torch.device(device).type == "cuda" else "torch", csr_ms=round((t1 - t0) * 1e3, 2))
    if torch.device(device).type == "cuda":
        C_row = spec["H"] * spec["f"]
        op.schedule_for(C_row if C_row >= 32 or spec["q"] == 1 else spec["q"] * C_row, True)
        sync()
        spec["operand_build"]["schedule_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
    return op, spec


def make_layer(op, spec, device):
    import tgcn_amd
    torch.manual_seed(1)
    if spec["cls"] == "TGCNCheb":
        layer = tgcn_amd.TGCNCheb(op, spec["f"], spec["g"], spec["K"])
    elif spec["cls"] == "TGCNCheb_H":
        layer = tgcn_amd.TGCNCheb_H(op, spec["f"], spec["g"], spec["K"], spec["H"])
    else:
        layer = tgcn_amd.GCNCheb(op, spec["f"], spec["g"], spec["K"])
    return layer.to(device)


def make_input(op, spec, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    q, n = spec["q"], op.n
    if spec["cls"] == "TGCNCheb":
        shape = (q, n, spec["f"])
    elif spec["cls"] == "TGCNCheb_H":
        shape = (q, n, spec["H"]) if spec["f"] == 1 else (q, n, spec["H"], spec["f"])
    else:
        shape = (q, n) if spec["f"] == 1 else (q, n, spec["f"])
    return torch.randn(shape, device=device, generator=g)


def measured_copy_gbps(device, nbytes=2 << 30):
    """device-to-device copy rate of this box (read + write bytes per second): the second denominator SURVEY.md 8(d) asks for"""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    best = 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def gather_ceiling(op, plan, C_row, device, fold=4096, reps=3):
    """The shipped hop_kernel on THIS operand's own rows, row lengths, entry order and schedule rules, with every column folded into `fold`
    rows of X spread evenly over the table (column c -> (c % fold) * (n_cols // fold)): all gathers then hit an L2-resident set (4096 rows of
    256 B = 1 MB per XCD), which is the most ANY locality scheme -- reordering, blocking, cache policy -- could make of this graph with this
    kernel.  One untimed launch, then `reps` timed ones (hipEvents of the library around the launch), outside the timed steps.
    -> (median ms per launch, fix-up ms)"""
    from tgcn_amd import _lib, functional as _F
    from tgcn_amd.graph import GraphOperand
    src = plan.rest if plan is not None else op            # the operand hops 2..K-1 run on (compact ids), or the plain one
    row, col, val = src.coo()
    n_cols = src.n_cols
    col = (col % fold) * max(1, n_cols // fold)
    fop = GraphOperand.from_coo(src.n, row, col, val, device, n_cols=n_cols)
    del row, col, val
    g = torch.Generator(device=device).manual_seed(99)
    x1 = torch.randn((1, n_cols, C_row), device=device, generator=g)
    y1 = torch.empty((1, src.n, C_row), device=device)
    _F.csr_hop(fop, x1, out=y1)
    torch.cuda.synchronize()
    _lib.profile_start(64)
    for _ in range(reps):
        _F.csr_hop(fop, x1, out=y1)
    torch.cuda.synchronize()
    prof = _lib.profile_stop(64)
    hop = sorted(ms for kind, ms in prof if kind == 0)
    fix = sorted(ms for kind, ms in prof if kind == 1)
    del fop, x1, y1
    torch.cuda.empty_cache()
    return hop[len(hop) // 2], (fix[len(fix) // 2] if fix else 0.0)


def scipy_baseline(op, spec, x, cols=8):
    """Single-thread scipy CSR, like the reference's numpy path (gcn/graph.py:256-265: Xt[k] = 2 L^k X - Xt[k-2] with
    scipy's .dot): the K-hop recursion of oracle.cheb_oracle.graph_chebyshev on `cols` of the C_in*H columns of ONE sample."""
    from oracle import cheb_oracle as O
    L = op.to_scipy().astype(np.float32)
    C_row = x[0].reshape(op.n, -1).shape[1]
    cols = min(cols, C_row)
    X = x[0].reshape(op.n, -1)[:, :cols].float().cpu().numpy()
    t0 = time.perf_counter()
    O.graph_chebyshev(L, X, spec["K"])
    dt = time.perf_counter() - t0
    units = op.nnz * (spec["K"] - 1) * spec["H"] * cols / C_row
    return dict(value=units / dt / 1e9, unit="G edge\u00b7timesteps/s", cores=1, kind="port",
                sample="%d of the %d columns of 1 of %d samples, K=%d recursion only (no projection), scipy CSR .dot single thread as gcn/graph.py:256-265, %.1f s" % (cols, C_row, spec["q"], spec["K"], dt))


def torch_dense_baseline(op, spec, layer, x, seconds=3.0):
    """SURVEY.md 8(d) baseline (ii), small graphs only: the reference's own evaluation order on the CPU -- torch einsum with the
    dense (n, n) operand -- through oracle.cheb_oracle.torch_dense_forward, all host threads torch uses by default."""
    from oracle import cheb_oracle as O
    Ld = torch.tensor(op.to_scipy().toarray(), dtype=torch.float32)
    horizon = spec["cls"] == "TGCNCheb_H"
    xc = x.float().cpu()
    if horizon:
        xc = xc.reshape(xc.shape[0], op.n, spec["H"], spec["f"])
    elif xc.dim() == 2:
        xc = xc.unsqueeze(-1)
    W, b = layer.weight.detach().cpu(), layer.bias.detach().cpu()
    O.torch_dense_forward(Ld, xc[:1], W, b, horizon)          # warm-up
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        out = O.torch_dense_forward(Ld, xc, W, b, horizon)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    units = op.nnz * (spec["K"] - 1) * spec["q"] * spec["H"]
    return out.numpy(), dict(value=units / dt / 1e9, unit="G edge\u00b7timesteps/s", cores=torch.get_num_threads(), kind="port",
                             sample="whole batch, dense (n, n) einsum as the reference evaluates it (gcn.py:72-78,147-153,230-236), torch CPU, %d runs of %.1f ms" % (reps, dt * 1e3))


def cpu_baseline(op, spec, layer, x, samples=(0,)):
    """oracle/cheb_ref.c (reference algorithm: full stack + unfolded weights) on the host cores, on the listed
    samples (time steps) of the same workload."""
    from oracle import c_port
    row, col, val = op.coo()
    rowptr = op.rowptr.cpu().numpy()
    col = col.to(torch.int32).cpu().numpy()
    val = val.cpu().numpy()
    samples = sorted(set(int(i) % spec["q"] for i in samples))
    q = len(samples)
    xs = np.stack([x[i].reshape(op.n, -1).float().cpu().numpy() for i in samples])
    K = spec["K"]
    W = layer.weight.detach().reshape(K, -1, spec["g"]).cpu().numpy()
    b = layer.bias.detach().reshape(-1).cpu().numpy()
    kind = 1 if spec["cls"] == "GCNCheb" else 2
    t0 = time.perf_counter()
    out = c_port.forward(0, rowptr, col, val, xs, W, b, kind)
    dt = time.perf_counter() - t0
    units = op.nnz * (K - 1) * q * spec["H"]
    return out, dict(value=units / dt / 1e9, unit="G edge\u00b7timesteps/s", cores=c_port.threads(), kind="port", seconds=dt, samples_timed=list(samples),
                     sample="samples %s of the %d of the same workload, full K=%d forward, oracle/cheb_ref.c OpenMP, %.1f s" % (samples, spec["q"], K, dt))


def host_description():
    """every CPU number is printed with the host it was measured on"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cpu_model=model, os_cpu_count=os.cpu_count(), torch_num_threads=torch.get_num_threads(), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS"),
                sched_affinity=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None)


MFMA_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3        # fp32 matrix peak


def other_workload_entry(name, device, steps=20, warmup=5):
    """One of BASELINE.json's other configurations (cfg2 f = 1 / 64, cfg3, cfg4; the headline's R-MAT with degree-sorted labels) in the driver's
    record: ms per forward, the roofline of ITS dominant kernel from the library's launch events, and the GPU result
    against the CPU restatements."""
    from tgcn_amd import _lib, functional as _F
    t_start = time.perf_counter()
    big = name.startswith("cfg5")
    if name == "cfg5_degree":
        # SURVEY.md 8(d) asks for BOTH labelings of the R-MAT; the headline is the random (worst-case) one, this is the degree-sorted ("friendly")
        # one -- which is the SLOWER of the two on this kernel (the hot rows share few L2 / memory channels)
        op, spec = build_workload("cfg5", "degree", device)
        steps, warmup = 3, 1
    else:
        op, spec = build_workload(name, "natural", device)
    layer = make_layer(op, spec, device)
    x = make_input(op, spec, device, seed=0)
    K, q, H = spec["K"], spec["q"], spec["H"]
    C_row, g_ch = H * spec["f"], spec["g"]
    with torch.no_grad():
        for _ in range(warmup):
            out = layer(x)
        torch.cuda.synchronize()
        # three rounds of `steps` forwards, the best one reported (all three kept in `rounds_ms`): these configurations take 40 ... 350 us per
        # forward, and the first round after the 150 GB of the headline were handed back to the driver has been seen 20 x slower than the next
        rounds = []
        for _ in range(1 if big else 3):
            out = None                     # (one output buffer at a time: 41 GB on the R-MAT)
            _lib.profile_start(8192)
            t0 = time.perf_counter()
            for _ in range(steps):
                if big:
                    out = None
                out = layer(x)
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0, _lib.profile_stop(8192)))
        dt, prof = min(rounds, key=lambda r: r[0])
    units = op.nnz * (K - 1) * q * H
    entry = dict(workload=spec["desc"], config=name, ms_per_step=round(dt / steps * 1e3, 4), value=round(units * steps / dt / 1e9, 3), unit="G edge\u00b7timesteps/s",
                 steps=steps, warmup=warmup, dtype="f32", rounds_ms=[round(r[0] / steps * 1e3, 4) for r in rounds])
    by_kind = {}
    for kind, ms in prof:
        by_kind.setdefault(kind, []).append(ms)
    names = {0: "hop_kernel", 1: "hop_fixup_kernel", 2: "projection", 3: "relayout_kernel", 4: "small_graph_one_launch"}
    entry["kernel_ms_per_step"] = {names.get(k, str(k)): round(float(np.sum(v)) / steps, 4) for k, v in sorted(by_kind.items())}
    dom = max(by_kind, key=lambda k: float(np.sum(by_kind[k]))) if by_kind else None
    pf = _F.use_project_first(q, op.n, C_row, g_ch) and not _F.small_path_tile(op, C_row, 0)
    F_cols = q * (g_ch if pf else C_row)
    bytes_recursion = (K - 1) * (8 * op.nnz + 4 * (op.n + 1) + 8 * op.n * F_cols)
    if dom == 4:
        layer_bytes = bytes_recursion + 4 * op.n * q * g_ch + 4 * K * C_row * g_ch + 4 * layer.bias.numel()
        mean_ms = float(np.mean(by_kind[4]))
        ach = layer_bytes / (mean_ms * 1e-3) / 1e9
        entry["roofline"] = dict(bound="hbm", kernel="small_forward_kernel / small_narrow_kernel (the one-launch small-graph kernels)", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4),
                                 traffic=None, algorithmic_bytes_per_launch=int(layer_bytes), mean_launch_ms=round(mean_ms, 4),
                                 launches_per_step=len(by_kind[4]) // steps,
                                 note="whole layer in one launch (CSR + activations in LDS): latency-bound configuration, BASELINE.md 4 sets no bar")
    elif dom == 2:
        mean_ms = float(np.sum(by_kind[2])) / steps
        flops = 2.0 * q * op.n * (K * C_row) * g_ch
        x3 = op.n * q >= 8192 and C_row * K >= 64
        peak = MFMA_BF16_PEAK_TFLOPS / 6 if x3 else MFMA_F32_PEAK_TFLOPS
        ach = flops / (mean_ms * 1e-3) / 1e12
        entry["roofline"] = dict(bound="mfma", kernel="project_x3v2_kernel (bf16x3)" if x3 else "project_kernel (fp32 mfma)", achieved=round(ach, 1), peak=round(peak, 1), unit="TFLOP/s",
                                 frac=round(ach / peak, 4), traffic=None, flops_per_step=int(flops), projection_ms_per_step=round(mean_ms, 4),
                                 note=("fp32-equivalent flops of the (K C_in H) x C_out contraction; peak = dense bf16 matrix peak / 6 -- every fp32 product is six bf16 "
                                       "MFMAs after the three-way operand split (DESIGN.md 3.2)") if x3 else "exact fp32 MFMA")
    elif dom == 0:
        n_l = len(by_kind[0]) // steps
        mean_ms = float(np.mean(by_kind[0]))
        ach = bytes_recursion / n_l / (mean_ms * 1e-3) / 1e9
        entry["roofline"] = dict(bound="hbm", kernel="hop_kernel", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4), traffic=None,
                                 algorithmic_bytes_per_launch=int(bytes_recursion / n_l), launches_per_step=n_l, mean_launch_ms=round(mean_ms, 4))
    if 0 in by_kind and dom != 0:
        n_l = len(by_kind[0]) // steps
        mean_ms = float(np.mean(by_kind[0]))
        entry["hop_roofline"] = dict(bound="hbm", achieved=round(bytes_recursion / n_l / (mean_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                     frac=round(bytes_recursion / n_l / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), launches_per_step=n_l, mean_launch_ms=round(mean_ms, 4),
                                     path="project-first (hops on C_out-wide rows)" if pf else "hops-first")
    check = [q - 1] if big else list(range(q))          # the R-MAT: one time step (the last: beyond 2^31 elements into x and out), ~11 s of CPU
    ref_out, cpu = cpu_baseline(op, spec, layer, x, samples=check)
    got = np.stack([out[i].cpu().numpy() for i in check])
    err = float(np.abs(got - ref_out).max() / np.abs(ref_out).max())
    entry["gpu_vs_cpu_rel_err"] = err
    entry["cpu_baseline"] = dict(value=round(cpu["value"], 4), unit=cpu["unit"], cores=cpu["cores"], kind="port", sample=cpu["sample"], seconds=round(cpu["seconds"], 3))
    assert err <= 1e-5, "%s: GPU result differs from the CPU restatement: %g" % (name, err)
    if op.n <= 4096 and spec["cls"] in ("GCNCheb", "TGCNCheb_H"):
        ref2, td = torch_dense_baseline(op, spec, layer, x, seconds=1.0)
        err2 = float(np.abs(out.cpu().numpy() - ref2).max() / np.abs(ref2).max())
        entry["torch_dense_einsum"] = dict(value=round(td["value"], 4), unit=td["unit"], cores=td["cores"], sample=td["sample"], gpu_vs_cpu_rel_err=err2)
        assert err2 <= 1e-5, "%s: GPU result differs from the dense-L einsum restatement: %g" % (name, err2)
    entry["seconds_in_bench"] = round(time.perf_counter() - t_start, 1)
    del layer, x, out, op
    torch.cuda.empty_cache()
    return entry


# ---- rehearsal compute (tests only): `--rehearsal-cpu` runs this file's N > 1 control flow (rendezvous, time sharding, extras,
# watchdog, the one JSON line) with the gloo backend on the CPU, the HIP calls replaced by scipy / numpy stand-ins that are
# injected through the hooks tgcn_amd/dist.py has for exactly this.  Nothing measured in this mode is a result (`data` says so);
# tests/test_bench_rehearsal.py drives it at world 2 under `pytest -m "not gpu"`.
class _CpuLayer:
    """reference_power layer on the CPU (tgcn/nn/gcn.py:66-78 + :39): stands in for the HIP module in rehearsals"""
    def __init__(self, op, spec):
        self.L = op.to_scipy()
        self.K = spec["K"]
        g = torch.Generator().manual_seed(1)
        self.W = (torch.rand(self.K, spec["H"] * spec["f"], spec["g"], generator=g) - 0.5).numpy()
        self.bias = None

    def __call__(self, x):
        q, n = x.shape[0], x.shape[1]
        xs = x.reshape(q, n, -1).numpy()
        Xt, P = [xs], xs
        for k in range(1, self.K):
            P = np.stack([self.L.dot(P[b]) for b in range(q)])
            Xt.append(P if k == 1 else 2 * P - Xt[k - 2])
        return torch.from_numpy(sum(Xt[k] @ self.W[k] for k in range(self.K)).astype(np.float32))


def build_sharded(op, spec, q_total, device, rank, world, mode, vertex_shards, rehearsal=False, exchange="auto"):
    """The vertex-sharded layer of SURVEY.md 8(e) on the same workload through its MODULE surface (tgcn_amd.dist.ShardedTGCNCheb /
    ShardedTGCNCheb_H: the reference's constructor arguments + a process group; weight fold, project-first and the exchange inside): every
    rank holds the seeded graph, owns an nnz-balanced row range (inside its group of `vertex_shards` ranks for mode "hybrid"; the groups
    split the q_total time steps, no communication between them) and its slice of x.
    -> (callable(overlap, qs), VertexShardedCheb, groups, time steps of this group, the module, this rank's x)"""
    from tgcn_amd import dist as tdist
    assert spec["cls"] in ("TGCNCheb", "TGCNCheb_H"), "vertex sharding bench is wired for the cfg5 / cfg4 layers"
    group, gi, ngroups = None, 0, 1
    if mode == "hybrid":
        group, gi, ngroups = tdist.hybrid_groups(world, vertex_shards)
    sl = tdist.shard_time_steps(q_total, gi, ngroups)
    q = sl.stop - sl.start
    row, col, val = op.coo()
    ops = None
    if rehearsal:
        from tools.cpu_standins import CpuOps
        ops = CpuOps()
    torch.manual_seed(1)
    graph = tdist.CooGraph(op.n, row, col, val)
    if spec["cls"] == "TGCNCheb":
        layer = tdist.ShardedTGCNCheb(graph, spec["f"], spec["g"], spec["K"], group=group, exchange=exchange, ops=ops)
    else:
        layer = tdist.ShardedTGCNCheb_H(graph, spec["f"], spec["g"], spec["K"], spec["H"], group=group, exchange=exchange, ops=ops)
    layer = layer.to(device)
    layer.force_sharded = True                     # also on one rank under a launcher: the exchange path meets the backend
    sh = layer.shard(device)                       # collective: partition, halo lists (tensor collectives), operands; parameters from the group's first rank
    layer.L = tdist.CooGraph(op.n, row[:0], col[:0], val[:0])      # the shard holds its rows: drop this rank's copy of the global entry list
    del row, col, val, graph
    C_in = spec["H"] * spec["f"]
    g = torch.Generator(device=device).manual_seed(rank)
    x_local = torch.randn((max(q, 1), sh.owned, C_in), device=device, generator=g)[:q]

    def fwd(overlap=True, qs=None):
        # qs: only the first qs of the group's time steps (the extras size themselves to their budget; columns are independent)
        layer.overlap = overlap
        return layer(x_local if qs is None else x_local[:qs])
    return fwd, sh, ngroups, q, layer, x_local


class Progress:
    """What the extras are doing right now, for the watchdog's record: the phase that is running, whether the host sits in a collective or in
    a device synchronise (and since when), and the seconds every finished phase took.  Plain attribute writes from the main thread, read once by
    the watchdog thread."""

    def __init__(self):
        self.t0 = time.perf_counter()
        self.phase, self.phase_t0 = None, self.t0
        self.in_sync_since = self.in_collective_since = None
        self.elapsed = []                      # [(phase, seconds)] in order

    def enter(self, phase):
        now = time.perf_counter()
        if self.phase is not None:
            self.elapsed.append((self.phase, round(now - self.phase_t0, 3)))
        self.phase, self.phase_t0 = phase, now

    def record(self, budget):
        now = time.perf_counter()
        sync_s = None if self.in_sync_since is None else now - self.in_sync_since
        coll_s = None if self.in_collective_since is None else now - self.in_collective_since
        # a run that is merely slow keeps finishing phases: every (sharding, form) pair sizes itself to a fraction of the budget, so ONE device
        # synchronise (or collective) that has lasted more than half of the whole budget is a stall, not a slow transport
        # (at least 20 s: a budget of a few seconds, as tests use, says nothing about a stall)
        stalled = max(sync_s or 0.0, coll_s or 0.0) > max(0.5 * budget, 20.0)
        return dict(phase=self.phase, phase_elapsed_s=round(now - self.phase_t0, 3), in_device_sync=sync_s is not None,
                    device_sync_elapsed_s=None if sync_s is None else round(sync_s, 3), in_collective=coll_s is not None,
                    collective_elapsed_s=None if coll_s is None else round(coll_s, 3), total_elapsed_s=round(now - self.t0, 3),
                    finished_phases_s=list(self.elapsed), stalled=stalled)


def run_extras(op, spec, q_total, device, rank, world, args, sync_all, dist, progress=None, out=None):
    """Vertex-sharded and hybrid runs of the same workload (2 timed forwards each), each in two forms: "plain" (one blocking
    exchange per hop, then the hop: the form with the fewest ways to go wrong on a first contact with RCCL) and "overlapped"
    (in-place receives, interior rows and other time steps under the exchange).  They must never cost the headline: every failure
    becomes an entry with an `error`, entries are appended to `out` as they finish, and every (sharding, form) pair SIZES ITSELF to
    its share of --extras-budget: the remaining budget is checked BEFORE anything runs (an entry whose share is already spent is skipped
    with an explicit error), one time step is run first (untimed: communicators, buffers), a second one is timed on every
    rank (max over ranks), and the measured forwards then take as many of the group's time steps as fit the share
    (`time_steps_used`; time steps are independent columns, so the rate per time step is the same quantity).  Should a run still
    overrun the whole budget, the caller's watchdog prints the headline with the entries finished so far, the phase record of `progress`,
    and ends every rank.  Every entry carries what each rank exchanges per hop (rows, bytes per channel and peer) and its per-phase times,
    so a slow or wrong run can be diagnosed from the one line."""
    out = [] if out is None else out
    progress = Progress() if progress is None else progress
    K, H = spec["K"], spec["H"]
    rehearsal = getattr(args, "rehearsal_cpu", False)
    modes = [("vertex", world)] + ([("hybrid", 2)] if world >= 4 and world % 2 == 0 else [])
    t_start = time.perf_counter()
    usable = 0.6 * args.extras_budget               # the rest: shard construction, the phase-log forward's extra syncs, slack
    n_entries = 2 * len(modes)

    def agree(v, red):
        t = torch.tensor([v], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        progress.in_collective_since = time.perf_counter()
        dist.all_reduce(t, op=red)
        v = float(t.item())
        progress.in_collective_since = None
        return v

    def synced():
        progress.in_sync_since = time.perf_counter()
        sync_all()
        progress.in_sync_since = None

    for mi, (mode, vs) in enumerate(modes):
        try:
            progress.enter("%s: building shards" % mode)
            fwd, sh, ngroups, qg, _, _ = build_sharded(op, dict(spec), q_total, device, rank, world, mode, vs, rehearsal, args.extras_exchange)
        except Exception as e:      # noqa: BLE001 -- reported, never fatal
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            out.append(dict(shard=mode, vertex_shards=vs, error="%s: %s" % (type(e).__name__, str(e)[:300])))
            continue
        for fi, (form, overlap) in enumerate((("plain", False), ("overlapped", True))):
            entry = dict(shard=mode, vertex_shards=vs, form=form)
            try:
                steps = 2
                sh.collect_stats = False
                progress.enter("%s / %s: budget check" % (mode, form))
                left = agree(usable - (time.perf_counter() - t_start), dist.ReduceOp.MIN)
                if left <= 0:      # shard construction (or an earlier entry) already ate the budget: do not even start the set-up forward
                    entry["error"] = "skipped: the usable extras budget (%.0f s of --extras-budget %.0f s) was spent before this entry's set-up forward" % (usable, args.extras_budget)
                    out.append(entry)
                    continue
                with torch.no_grad():
                    progress.enter("%s / %s: set-up forward (1 time step, untimed)" % (mode, form))
                    fwd(overlap, 1)                         # set-up: communicators, exchange buffers, allocator
                    synced()
                    progress.enter("%s / %s: one timed time step" % (mode, form))
                    t0 = time.perf_counter()
                    fwd(overlap, 1)
                    synced()
                    t1 = agree(time.perf_counter() - t0, dist.ReduceOp.MAX)          # one time step, slowest rank
                    # this entry's share of what is left of the usable budget, spread over its steps + 1 phase-log forwards;
                    # every rank computes the same number from the same all-reduced inputs
                    left = agree(usable - (time.perf_counter() - t_start), dist.ReduceOp.MIN)
                    share = max(left, 0.0) / max(1, n_entries - (2 * mi + fi))
                    qs = int(max(1, min(qg, share / ((steps + 1.5) * max(t1, 1e-6)))))
                    # a transport on which even ONE time step eats the share (gloo through the host in rehearsals; a slow first contact with
                    # RCCL): fewer forwards -- one measured step, and below that the timed one-step forward above IS the measurement
                    if share < 2.5 * t1:
                        steps, dt, qs, phases = 1, t1, 1, None
                    else:
                        if share < (steps + 1.5) * t1:
                            steps = 1
                        progress.enter("%s / %s: %d measured forward(s) of %d time step(s)" % (mode, form, steps, qs))
                        t0 = time.perf_counter()
                        for _ in range(steps):
                            fwd(overlap, qs)
                        synced()
                        dt = time.perf_counter() - t0
                        progress.enter("%s / %s: phase-log forward" % (mode, form))
                        sh.collect_stats = True                # one more forward with the phase log (device events add their own syncs)
                        fwd(overlap, qs)
                        synced()
                        sh.collect_stats = False
                        phases = sh.stats
                        dt = agree(dt, dist.ReduceOp.MAX)
                progress.enter("%s / %s: gathering the per-rank records" % (mode, form))
                mine = dict(sh.describe(), phases_ms=phases)
                per_rank = [None] * world
                progress.in_collective_since = time.perf_counter()
                dist.all_gather_object(per_rank, mine)
                progress.in_collective_since = None
                # the groups of a hybrid grid run side by side: together they cover ngroups * qs time steps per forward
                entry.update(value=round(op.nnz * (K - 1) * (qs * ngroups) * H * steps / dt / 1e9, 3), unit="G edge·timesteps/s", ms_per_step=round(dt / steps * 1e3, 3),
                             steps=steps, scaling="strong", exchange=sh.exchange, groups=ngroups, time_steps_per_group=qg, time_steps_used=qs,
                             one_time_step_ms=round(t1 * 1e3, 3),
                             hop_row_floats=mine["row_floats"], message_bytes_per_hop_and_time_step_rank0=mine["bytes_in_per_hop_and_time_step"], ranks=per_rank)
            except Exception as e:      # noqa: BLE001 -- reported, never fatal
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                entry["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
            progress.in_sync_since = progress.in_collective_since = None
            out.append(entry)
        del fwd, sh
        if device.type == "cuda":
            torch.cuda.empty_cache()
    progress.enter("done")
    return out


def count_gpus_without_runtime():
    """GPUs this process may use, without a HIP call: the agents of /sys/class/kfd/kfd/topology/nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set.  0 where the driver's topology directory does
    not exist (no amdgpu compute driver), None when it exists but cannot be parsed."""
    import glob
    n = 0
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        return 0                                   # no amdgpu compute driver on this host: no GPU a ROCm process could open
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    try:
        for path in files:
            props = dict(ln.split()[:2] for ln in open(path) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: this process becomes the parent of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` -- a CHILD process, never an exec -- and
    returns its exit code; the ranks inherit stdout / stderr, so rank 0's ONE JSON line is this command's output.  Nothing here touches
    the GPU: the devices are counted from the kernel driver's topology files (count_gpus_without_runtime: no HIP call -- torch.cuda.device_count()
    can fall back to hipGetDeviceCount, which initialises the runtime in this parent), and a rank count the node cannot hold is refused before
    anything starts instead of being run on fewer GPUs; where the count cannot be read the ranks themselves fail on set_device."""
    import socket
    import subprocess
    if not args.rehearsal_cpu:
        have = count_gpus_without_runtime()
        if have is not None and args.gpus > have:
            print("bench.py: --gpus %d but this node has %d GPU(s); refusing to run the scaling bench on fewer devices" % (args.gpus, have), file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool: RCCL's cross-process handles need it
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    import signal
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True)      # its own process group: the ranks can be ended with it

    def forward(signum, _frame):          # a parent that is told to stop takes its ranks with it (exact process group, never a pattern)
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass
    old = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}
    try:
        return child.wait()
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg5")
    ap.add_argument("--labeling", default="random")
    ap.add_argument("--vertices", type=int, default=None, help="override vertex count (cfg5 only; reported in config)")
    ap.add_argument("--entries", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=100.0, help="seconds of CPU-baseline time on the headline workload: time steps are timed two at a time until this is used up (all 16 when it allows); the rest is scaled and said so")
    ap.add_argument("--no-others", action="store_true", help="default run (cfg5, 1 GPU): skip `other_workloads` (cfg2 f=1 / f=64, cfg3, cfg4 measured after the headline)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip roofline.gather_ceiling_ms (one extra operand build + 4 hop launches after the timed steps)")
    ap.add_argument("--shard", default="time", choices=["time", "vertex", "hybrid"], help="N > 1: time steps split over the ranks (no collective); vertex rows per rank with a halo / all-gather exchange per hop; hybrid: --vertex-shards ranks share a graph, groups split the time steps")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1, time sharding: strong = the workload's q time steps split over the ranks (default); weak = q time steps per rank")
    ap.add_argument("--vertex-shards", type=int, default=2, help="ranks per graph copy for --shard hybrid")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--project-variant", type=int, default=None, help="developer: tgcn_set_tuning(project_variant, v)")
    ap.add_argument("--no-small-path", action="store_true", help="developer: disable the one-launch small-graph kernel")
    ap.add_argument("--compact-q-chunk", type=int, default=None, help="developer: time steps per pass of the compacted forward")
    ap.add_argument("--no-compact", action="store_true", help="developer: hop tensors for all vertices even when many rows are empty")
    ap.add_argument("--tune", action="append", default=[], help="developer: key=value for tgcn_set_tuning (repeatable)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1, time sharding: skip the vertex-sharded / hybrid runs reported in `other_shardings`")
    ap.add_argument("--rehearsal-cpu", action="store_true", help="tests only: run the N > 1 control flow on the CPU with the gloo backend and scipy stand-ins for the HIP calls; nothing measured in this mode is a result")
    ap.add_argument("--force-extras", action="store_true", help="run the vertex-sharded extras also at world 1 (one rank under a launcher: the collectives of the N > 1 path meet RCCL)")
    ap.add_argument("--extras-exchange", default="auto", choices=["auto", "halo", "allgather"], help="exchange form of the vertex-sharded runs (auto: all-gather when the halo is most of the graph)")
    ap.add_argument("--extras-budget", type=float, default=150.0, help="seconds after which the extra runs are abandoned and the headline line is printed without them")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # started by torch.distributed.run (the driver's N > 1 command, or self_launch)
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))          # parent: no GPU call has been made in this process, and none will be
    if not launched and args.shard in ("vertex", "hybrid"):
        # `python bench.py --workload cfg4 --shard vertex` on one GPU: the sharded layer needs a process group also at world size 1 (its
        # collectives run on the real backend with one rank) -- this process becomes that rank instead of quietly measuring the single-GPU driver
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        s_.close()
        launched = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = args.rehearsal_cpu
    if rehearsal:
        args.backend, args.no_cpu = "gloo", True
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        local = local % torch.cuda.device_count()      # rehearsals may put several ranks on one card
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with torch.distributed.run --nproc-per-node %d (or plain `python bench.py --gpus %d`, "
                         "which starts that launcher itself)" % (args.gpus, world, args.gpus, args.gpus))
    dist = None
    if launched:          # also at world 1: `torch.distributed.run --nproc-per-node 1 bench.py` runs the collectives of the N > 1 path on one rank
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from tgcn_amd import _lib, functional as _F
    if args.no_small_path:
        _F.SMALL_PATH = False
    _F.COMPACT_Q_CHUNK = args.compact_q_chunk
    if args.no_compact:
        _F.COMPACT = False
    if args.project_variant is not None:
        _lib.check(_lib.lib().tgcn_set_tuning(b"project_variant", args.project_variant))
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.lib().tgcn_set_tuning(k.encode(), int(v)))
    op, spec = build_workload(args.workload, args.labeling, device, args.vertices, args.entries)
    q_total = spec["q"]
    strong_time = world > 1 and args.shard == "time" and args.scaling == "strong" and spec["q"] >= world
    if strong_time:
        from tgcn_amd.dist import shard_time_steps
        mine = shard_time_steps(spec["q"], rank, world)
        spec["q"] = mine.stop - mine.start
    K, q, H = spec["K"], spec["q"], spec["H"]
    vertex_mode = dist is not None and args.shard in ("vertex", "hybrid")       # world 1 under a launcher: the same code on one rank (first contact with RCCL)
    ngroups = 1
    if vertex_mode:
        fwd, sh, ngroups, q, layer, x = build_sharded(op, spec, q_total, device, rank, world, args.shard, args.vertex_shards, rehearsal, args.extras_exchange)
        spec["q"] = q          # the module itself is stepped: layer(x_local) -> out_local (overlapped form); at world 1 its rows are the whole graph,
                               # so the CPU leg below checks it against the oracle like the single-GPU layer
    elif rehearsal:
        layer = _CpuLayer(op, spec)
        x = make_input(op, spec, device, seed=rank)
    else:
        layer = make_layer(op, spec, device)
        x = make_input(op, spec, device, seed=rank)

    def sync_all():
        if device.type == "cuda":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if device.type == "cuda":
                torch.cuda.synchronize()

    with torch.no_grad():
        # set-up, not a step: one forward so that the caching allocator owns the output / workspace blocks (a cold
        # hipMalloc of the 41 GB cfg5 output takes ~1 s); `out = None` first keeps it at ONE output buffer, otherwise the
        # second forward would allocate its result while the first is still referenced
        out = layer(x)
        for _ in range(args.warmup):
            out = None
            out = layer(x)
        sync_all()
        if not rehearsal:
            _lib.profile_start(65536)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = None
            out = layer(x)
        sync_all()
        dt = time.perf_counter() - t0
        prof = [] if rehearsal else _lib.profile_stop(65536)
    per_rank = None
    if dist is not None:
        # what every rank measured on its own (the headline is the MAX): a scaling run that comes out below N x is read from these -- a slow rank,
        # a rank with more time steps, a rank whose hop launches are slower -- without a second run
        mine = dict(rank=rank, time_steps=spec["q"] * spec["H"], ms_per_step=round(dt / args.steps * 1e3, 3),
                    hop_mean_launch_ms=(round(float(np.mean([ms for kind, ms in prof if kind == 0])), 4) if any(kind == 0 for kind, _ in prof) else None),
                    device=(torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        tmax = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    units_per_step = op.nnz * (K - 1) * q * H            # edge.timesteps per forward per rank
    if strong_time or vertex_mode:
        value = op.nnz * (K - 1) * q_total * H * args.steps / dt / 1e9                      # the ranks split ONE q_total-step forward
    else:
        value = world * units_per_step * args.steps / dt / 1e9

    # ---- roofline of the dominant kernel (hop_kernel): algorithmic bytes per launch / mean launch duration
    hop_ms = [ms for kind, ms in prof if kind == 0]            # full hop launches (every row of the operand)
    hop_long_ms = [ms for kind, ms in prof if kind == 7]       # last hop fused into the projection: the launch covers the rows above the threshold only
    proj_ms = [ms for kind, ms in prof if kind in (2, 8)]      # 8: the projection that gathers the last hop's short rows itself
    fix_ms = [ms for kind, ms in prof if kind == 1]
    C_row = H * spec["f"]
    F = q * C_row
    if vertex_mode:
        pf_path = bool(hop_ms) and sh.use_project_first(C_row, spec["g"], K)
    else:
        pf_path = hop_ms and _F.use_project_first(q, op.n, C_row, spec["g"]) and not _F.small_path_tile(op, C_row, 0)
    if pf_path:
        F = q * spec["g"]      # project-first: the hops run on the (q, n, C_out) results, not on the C_in*H-wide inputs
    nnz_l, n_l = (sh.op.nnz, sh.owned) if vertex_mode else (op.nnz, op.n)      # what ONE rank's launches process
    bytes_recursion = (K - 1) * (8 * nnz_l + 4 * (n_l + 1) + 8 * n_l * F)
    n_hop_launches = (len(hop_ms) + len(hop_long_ms)) // args.steps if hop_ms else 0      # hops of one forward, fused or not
    roofline = None
    plan_r = None
    small_ms = [ms for kind, ms in prof if kind == 4]
    if small_ms and not hop_ms:
        # one-launch LDS-resident path: the whole layer is one kernel; algorithmic bytes = SURVEY 8(d) whole-layer figure
        bias_elems = layer.bias.numel() if layer.bias is not None else 0
        layer_bytes = bytes_recursion + 4 * op.n * q * spec["g"] + 4 * K * C_row * spec["g"] + 4 * bias_elems
        mean_ms = float(np.mean(small_ms))
        achieved = layer_bytes / (mean_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="small_forward_kernel", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=None, algorithmic_bytes_per_launch=int(layer_bytes),
                        launches_per_step=len(small_ms) // args.steps, mean_launch_ms=round(mean_ms, 4),
                        note="whole layer in one launch; hop tensors never leave LDS, so the HBM roofline on recursion bytes is nominal")
    if hop_ms:
        bytes_per_launch = bytes_recursion / n_hop_launches
        mean_ms = float(np.mean(hop_ms))
        achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9
        # HBM-side bytes per hop from the rocprofv3 --pmc passes of THIS code (tools/collect_traffic.sh writes the file with
        # the hash of the kernel sources it profiled); a file taken from other sources is stale and reported as null
        traffic = None
        traffic_note = "no counter file for this workload"
        tpath = ""
        if args.vertices is None and args.entries is None and not vertex_mode:
            tpath = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (args.workload, args.labeling))
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            have = _lib.binary_hash()          # the stamp of the library that is loaded, not of the sources on disk
            if tj.get("source_hash") == have:
                traffic = tj.get("hbm_bytes_per_hop_launch")
                traffic_note = "%s (source hash %s)" % (os.path.relpath(tpath, ROOT), tj.get("source_hash"))
            else:
                traffic_note = "%s is from other kernel sources (%s, loaded binary %s): stale, not reported" % (os.path.relpath(tpath, ROOT), tj.get("source_hash"), have)
        copy_gbps = measured_copy_gbps(device)
        roofline = dict(bound="hbm", kernel="hop_kernel", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_note,
                        copy_peak_measured=round(copy_gbps, 1), frac_of_copy_peak=round(achieved / copy_gbps, 4),
                        algorithmic_bytes_per_launch=int(bytes_per_launch), launches_per_step=n_hop_launches,
                        path="project-first (hops on C_out-wide rows)" if pf_path else "hops-first",
                        mean_launch_ms=round(mean_ms, 4), hop_ms_per_step=round(float(np.sum(hop_ms)) / args.steps, 3),
                        full_hop_launches_per_step=len(hop_ms) // args.steps,
                        hop_long_rows_only_ms_per_step=round(float(np.sum(hop_long_ms)) / args.steps, 3),
                        fixup_ms_per_step=round(float(np.sum(fix_ms)) / args.steps, 3),
                        project_ms_per_step=round(float(np.sum(proj_ms)) / args.steps, 3))
        # what one hop launch PHYSICALLY has to move once: every stored entry, the row pointers, and each row that is
        # touched once in and once out -- on the compacted operand only the vertices with entries exist as rows, so this is less than the
        # SURVEY 8(d) figure above (which prices all n rows): a compaction gain is fewer rows processed, not more bandwidth
        plan_r = None if (vertex_mode or pf_path or small_ms) else (op.compact_plan() if (_F.COMPACT and 2 <= K <= 32 and _F.choose_layout(q, op.n, C_row) == 0) else None)
        n_rows = plan_r.n_c if plan_r is not None else n_l
        launches_per_hop = max(1, n_hop_launches // max(1, K - 1))            # cfg5: one launch per hop and time step
        phys = 8 * nnz_l + 4 * (n_rows + 1) + 8 * n_rows * (F / launches_per_hop)
        roofline["physical_bytes_per_launch"] = int(phys)
        roofline["physical_achieved"] = round(phys / (mean_ms * 1e-3) / 1e9, 1)
        roofline["physical_frac"] = round(phys / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        roofline["physical_note"] = ("entries + row pointers + each of the %d rows that exist in the hop tensors once in and once out; the entry stream is "
                                     "read by every launch" % n_rows)
        if not hop_long_ms and n_hop_launches % max(1, K - 1) == 0 and K > 2:
            # mean launch time by hop position (hop 1 gathers from x in the caller's labels, the others from compact hop tensors): the launches of a
            # time step follow each other in hop order on every path of the layer driver
            per = np.array(hop_ms[: (len(hop_ms) // (K - 1)) * (K - 1)]).reshape(-1, K - 1).mean(axis=0)
            roofline["mean_launch_ms_by_hop"] = [round(float(v), 4) for v in per]
        if hop_long_ms:
            roofline["note"] = ("last hop fused into the projection: %d of the %d hop launches per step cover the rows above the threshold only and are NOT in "
                                "mean_launch_ms; frac is for the full hop launches" % (len(hop_long_ms) // args.steps, n_hop_launches))
        if world == 1 and not vertex_mode and not pf_path and not small_ms and args.workload in ("cfg5", "cfg5n") and not args.no_ceiling:
            # how far the measured launch is from the best any locality scheme could reach, in the driver's record
            try:
                layout1 = _F.choose_layout(q, op.n, C_row) == 1
                ceil_ms, ceil_fix = gather_ceiling(op, plan_r, q * C_row if layout1 else C_row, device)
                roofline["gather_ceiling_ms"] = round(ceil_ms, 4)
                roofline["frac_of_gather_ceiling"] = round(ceil_ms / mean_ms, 4)
                roofline["gather_ceiling_note"] = ("the same hop_kernel on the same rows / row lengths / schedule rules with every column folded into 4096 rows of X "
                                                   "(all gathers L2 hits): the bound of ANY locality scheme for this kernel on this graph; measured after the timed "
                                                   "steps, 1 untimed + 3 timed launches (median); its fix-up %.3f ms" % ceil_fix)
            except Exception as e:      # noqa: BLE001 -- a diagnostic never costs the headline
                roofline["gather_ceiling_ms"] = None
                roofline["gather_ceiling_note"] = "not measured: %s: %s" % (type(e).__name__, str(e)[:200])
        step_s = dt / args.steps
        roofline["whole_step"] = dict(bytes=int(bytes_recursion), achieved=round(bytes_recursion / step_s / 1e9, 1), unit="GB/s",
                                      frac=round(bytes_recursion / step_s / 1e9 / HBM_PEAK_GBPS, 4),
                                      note="SURVEY 8(d) recursion bytes of one rank's forward / its whole forward time (hops + fix-up + projection)")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not (vertex_mode and rehearsal):
        # first AND last sample: the last one sits behind (q-1)*n*C elements (> 2^31 on cfg5), so the check also covers the
        # 64-bit address arithmetic of the run that was timed
        check = [0, q - 1] if (q > 1 and op.n * C_row * q >= 2 ** 31) else [0]
        ref_out, cpu = cpu_baseline(op, spec, layer, x, samples=check)
        got = np.stack([out[i].cpu().numpy() for i in check])
        errs = [float(np.abs(got[i] - ref_out[i]).max() / np.abs(ref_out[i]).max()) for i in range(len(check))]
        # more time steps, two at a time, while --cpu-budget lasts ("executed per time-step chunk and summed"); every one is also a parity check
        timed, spent, units_done = list(check), cpu["seconds"], cpu["value"] * 1e9 * cpu["seconds"]
        rest = [i for i in range(q) if i not in check]
        # (chunks of two time steps on the headline, where a step takes ~10 s and 25 GB of host memory; the small workloads take the rest in one call)
        per_call = 2 if cpu["seconds"] / len(check) > 0.5 else len(rest)
        while rest and spent + 1.15 * (spent / len(timed)) * min(per_call, len(rest)) <= args.cpu_budget:
            chunk, rest = rest[:per_call], rest[per_call:]
            ref_c, c2 = cpu_baseline(op, spec, layer, x, samples=chunk)
            errs += [float(np.abs(out[i].cpu().numpy() - ref_c[j]).max() / np.abs(ref_c[j]).max()) for j, i in enumerate(chunk)]
            timed += chunk
            spent += c2["seconds"]
            units_done += c2["value"] * 1e9 * c2["seconds"]
            del ref_c
        cpu.update(value=units_done / max(spent, 1e-9) / 1e9, seconds=round(spent, 3), samples_timed=sorted(timed),
                   scaled_from=None if len(timed) == q else "%d/%d" % (len(timed), q),
                   sample="time steps %s of the %d of the same workload (two per call, summed), full K=%d forward, oracle/cheb_ref.c OpenMP, %.1f s%s"
                          % (sorted(timed), q, K, spent, "" if len(timed) == q else "; the rate is per time step, so the other %d scale 1:1 (--cpu-budget %.0f s)" % (q - len(timed), args.cpu_budget)))
        cpu["host"] = host_description()
        check = sorted(timed)
        err = max(errs)
        cpu["gpu_vs_cpu_rel_err"] = err
        cpu["checked_samples"] = check
        assert err <= 1e-5, "GPU result differs from the CPU restatement: %s" % errs
        cpu["scipy_single_thread"] = scipy_baseline(op, spec, x)
        if op.n <= 4096 and spec["cls"] in ("GCNCheb", "TGCNCheb_H"):
            ref2, cpu["torch_dense_einsum"] = torch_dense_baseline(op, spec, layer, x)
            err2 = float(np.abs(out.cpu().numpy() - ref2).max() / np.abs(ref2).max())
            cpu["torch_dense_einsum"]["gpu_vs_cpu_rel_err"] = err2
            assert err2 <= 1e-5, "GPU result differs from the dense-L einsum restatement: %g" % err2

    # what the arithmetic of this run was (the label of `dtype`): hops and accumulators are fp32 throughout; large projections
    # take the bf16 matrix pipe with every fp32 operand split three ways (fp32-equivalent)
    plan = None if vertex_mode else (op.compact_plan() if (_F.COMPACT and spec["cls"] in ("TGCNCheb", "TGCNCheb_H", "GCNCheb") and not small_ms and not pf_path
                                                           and 2 <= K <= 32 and _F.choose_layout(q, op.n, C_row) == 0) else None)
    x3_proj = (not small_ms) and op.n * max(q, 1) >= 8192 and C_row * K >= 64 and (args.project_variant in (None, 0, 3))
    line = None
    if rank == 0:
        line = dict(metric=baseline_metric(), value=round(value, 3), unit="G edge\u00b7timesteps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling="strong" if (vertex_mode or strong_time or world == 1) else "weak", vs_baseline=None,
                    dtype="f32", data="rehearsal on the CPU with scipy stand-ins: control flow only, NOT a measurement" if rehearsal else "synthetic",
                    config=dict(workload=spec["desc"], dist_backend=(args.backend if dist is not None else None), K=K, time_steps_per_gpu=q * H, C_in=spec["f"], C_out=spec["g"],
                                sharding=(("%d group(s) x %d vertex shards, %s exchange per hop inside a group" % (ngroups, sh.world, sh.exchange)) if ngroups > 1 else ("vertex rows across ranks, %s exchange per hop" % sh.exchange)) if vertex_mode else (("the %d time steps of the workload split over %d ranks (%d on rank 0), CSR replicated, no collective" % (q_total, world, q) if strong_time else "%d time steps per rank, CSR replicated, no collective" % q) if world > 1 else "single GPU"),
                                nnz=op.nnz, n=op.n,
                                arithmetic="f32 (bf16x3 projection: fp32 operands split into three bf16 terms on the matrix pipe)" if x3_proj else "f32",
                                operand_build=spec.get("operand_build"),
                                empty_rows=(plan.n_empty if plan is not None else 0),
                                hop_tensors="compact (%d of %d vertices have stored entries)" % (plan.n_c, op.n) if plan is not None else "all vertices",
                                time_steps_per_pass=(sorted(set(plan.q_chunk_cache.values())) if plan is not None and plan.q_chunk_cache else None)),
                    roofline=roofline, cpu_baseline=cpu)
        if per_rank is not None:
            line["ranks"] = per_rank
    # ---- N > 1: the mandated vertex-sharded scheme (and the hybrid grid) on the same workload, reported next to the headline.
    # A watchdog on every rank prints the headline without them and ends the process if they overrun their budget.
    if dist is not None and (world > 1 or args.force_extras) and args.shard == "time" and not args.no_extras and spec["cls"] in ("TGCNCheb", "TGCNCheb_H"):
        import threading
        printed = threading.Lock()         # exactly one JSON line, whoever gets there first
        progress = Progress()
        extras = []                        # entries finished so far: the watchdog prints them with the headline

        def bail():
            # the extras overran --extras-budget.  The headline WAS measured, so print it -- marked `extras_abandoned`, with the entries that
            # finished and the machine-readable record of where the extras stood (`extras_abandon`: phase, whether the host sat in a device
            # synchronise or a collective and for how long, seconds per finished phase).  Exit code: 0 when the run was merely slow (phases kept
            # finishing); 3 when ONE device synchronise / collective had been pending for more than half of the whole budget -- a stalled
            # exchange or a GPU hang must not read as success on the 8-GPU box.  Plain exit, never a re-exec; a rank stuck in a collective
            # cannot be joined, hence os._exit.
            if not printed.acquire(blocking=False):
                return
            rec = progress.record(args.extras_budget)
            if rank == 0:
                line["extras_abandoned"] = True
                line["extras_abandon"] = rec
                line["other_shardings"] = list(extras) + [dict(error="abandoned after %.0f s (--extras-budget)" % args.extras_budget, last_started=rec["phase"])]
                print(json.dumps(line), flush=True)
            sys.stdout.flush()
            os._exit(3 if rec["stalled"] else 0)
        dog = threading.Timer(args.extras_budget, bail)
        dog.daemon = True
        dog.start()
        run_extras(op, spec, q_total, device, rank, world, args, sync_all, dist, progress, extras)
        dog.cancel()
        if not printed.acquire(blocking=False):      # the watchdog fired while the extras were returning: it prints and exits
            time.sleep(3600)
        if rank == 0:
            line["other_shardings"] = extras
            print(json.dumps(line), flush=True)
    elif rank == 0:
        if world == 1 and dist is None and args.workload == "cfg5" and not args.no_others and not rehearsal and args.vertices is None and args.entries is None:
            # the driver's one command measures BASELINE.json's other configurations too, after the headline and with
            # everything of the headline freed; the headline's fields above are final at this point.  A failure here is reported, never fatal.
            del out, x, layer, op
            plan = plan_r = None
            torch.cuda.empty_cache()
            others = []
            for name in ("cfg5_degree", "cfg2", "cfg2w", "cfg3", "cfg4"):
                try:
                    others.append(other_workload_entry(name, device))
                except Exception as e:      # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    others.append(dict(config=name, error="%s: %s" % (type(e).__name__, str(e)[:300])))
            line["other_workloads"] = others
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
