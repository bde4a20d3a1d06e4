"""bench.py's own N > 1 control flow at world 2 on the CPU (gloo): rendezvous from the launcher's environment, strong time
sharding, the vertex-sharded extras (plain + overlapped forms with their per-rank diagnostics), the watchdog, and the ONE JSON
line.  The HIP calls are replaced by the scipy stand-ins of `bench.py --rehearsal-cpu` (injected through tgcn_amd/dist.py's
hooks); nothing here is a measurement.  The multi-GPU runs themselves belong to the driver."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(extra, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rehearsal-cpu", "--vertices", "3000", "--entries", "40000"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_strong_time_sharding_and_extras_at_world_2():
    r = _launch([])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "split over 2 ranks (8 on rank 0)" in line["config"]["sharding"]
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and all(r["time_steps"] == 8 and r["ms_per_step"] > 0 for r in line["ranks"])
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in line["ranks"]) - 1e-3          # the headline is the slowest rank
    assert "extras_abandoned" not in line
    extras = line["other_shardings"]
    assert [(e["shard"], e["form"]) for e in extras] == [("vertex", "plain"), ("vertex", "overlapped")]
    for e in extras:
        assert "error" not in e, e
        assert e["value"] > 0 and e["exchange"] in ("halo", "allgather") and len(e["ranks"]) == 2 and e["time_steps_used"] >= 1
        for rk in e["ranks"]:           # what a slow or wrong RCCL run would be diagnosed from
            assert rk["owned_rows"] > 0 and rk["bytes_per_channel_in"] > 0 and rk["phases_ms"]
            assert any(k.startswith("exchange") for k in rk["phases_ms"])
        assert sum(rk["owned_rows"] for rk in e["ranks"]) == 3000


def test_abandoned_extras_print_the_headline_and_still_exit_zero():
    """the time-sharded headline exits 0 whenever it was measured -- also when the extras run out of budget."""
    r = _launch(["--extras-budget", "0.001"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["extras_abandoned"] is True and lines[0]["value"] > 0
    assert "abandoned" in lines[0]["other_shardings"][-1]["error"]


def test_extras_size_themselves_to_a_small_budget():
    """A budget that does not hold all 8 time steps of a group: the entries still finish (fewer time steps, recorded), rc 0."""
    r = _launch(["--extras-budget", "20", "--vertices", "20000", "--entries", "300000"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_lines(r.stdout)[0]
    assert "extras_abandoned" not in line
    extras = line["other_shardings"]
    assert [(e["shard"], e["form"]) for e in extras] == [("vertex", "plain"), ("vertex", "overlapped")]
    for e in extras:
        assert "error" not in e, e
        assert 1 <= e["time_steps_used"] <= e["time_steps_per_group"] == 16 and e["value"] > 0 and e["one_time_step_ms"] > 0


def _bare(extra, env_extra=None, timeout=300):
    """`python bench.py ...` with NO launcher and no rendezvous variables in the environment"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` used to run ONE rank and print n_gpus 1 when WORLD_SIZE was unset; the parent now
    starts torch.distributed.run as a child and relays the one line and the exit code."""
    r = _bare(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearsal-cpu", "--vertices", "3000", "--entries", "40000", "--no-extras"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["dist_backend"] == "gloo"
    assert "split over 2 ranks" in lines[0]["config"]["sharding"]
    assert "starting 2 ranks" in r.stderr


def test_more_ranks_than_gpus_is_refused_not_degraded():
    """no GPU in the CPU suite's container (or fewer than 64 anywhere): the scaling command must fail, not run on fewer devices"""
    r = _bare(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing" in r.stderr and not _json_lines(r.stdout)


def test_rank_count_mismatch_with_the_launcher_is_an_error():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal-cpu", "--vertices", "3000", "--entries", "40000"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not _json_lines(r.stdout)


def test_one_rank_under_the_launcher_runs_the_collectives_and_the_extras():
    """world 1 with a process group: the all-reduce of the timing and (--force-extras) the all-gather form of the vertex-sharded layer run
    on one rank -- the CPU twin of the GPU suite's first contact with RCCL (tests/test_dist_gpu.py)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rehearsal-cpu",
           "--vertices", "3000", "--entries", "40000", "--force-extras", "--extras-exchange", "allgather"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_lines(r.stdout)[0]
    assert line["n_gpus"] == 1 and line["config"]["dist_backend"] == "gloo"
    assert [(e["shard"], e["form"], e.get("exchange")) for e in line["other_shardings"]] == [("vertex", "plain", "allgather"), ("vertex", "overlapped", "allgather")]


def test_the_drivers_eight_rank_command_on_the_cpu():
    """The command an 8-GPU lease runs (`--gpus 8`, here self-launched, gloo, scipy stand-ins): 16 time steps split 2 per rank, the vertex-sharded
    extras over all 8 ranks and the hybrid grid (4 groups x 2 vertex shards), one line, rc 0.  Control flow only -- nothing here is a measurement."""
    r = _bare(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rehearsal-cpu", "--vertices", "4000", "--entries", "50000", "--extras-budget", "240"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and "split over 8 ranks (2 on rank 0)" in line["config"]["sharding"]
    assert "extras_abandoned" not in line, line.get("extras_abandon")
    got = [(e["shard"], e["vertex_shards"], e["form"]) for e in line["other_shardings"]]
    assert got == [("vertex", 8, "plain"), ("vertex", 8, "overlapped"), ("hybrid", 2, "plain"), ("hybrid", 2, "overlapped")], got
    for e in line["other_shardings"]:
        assert "error" not in e, e
        assert len(e["ranks"]) == 8 and e["value"] > 0
        if e["shard"] == "hybrid":
            assert e["groups"] == 4 and e["time_steps_per_group"] == 4
            assert sum(rk["owned_rows"] for rk in e["ranks"]) == 4 * 4000          # every group holds the whole graph
        else:
            assert sum(rk["owned_rows"] for rk in e["ranks"]) == 4000


def test_shard_vertex_without_a_launcher_is_the_sharded_layer_on_one_rank():
    """`python bench.py --workload cfg4 --shard vertex` used to fall through to the single-GPU driver when no launcher had set
    WORLD_SIZE: the process now forms a one-rank group itself and runs the vertex-sharded MODULE (here: CPU rehearsal, gloo, stand-in arithmetic)"""
    r = _bare(["--rehearsal-cpu", "--workload", "cfg4", "--shard", "vertex", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1
    assert lines[0]["config"]["sharding"].startswith("vertex rows across ranks") and lines[0]["config"]["dist_backend"] == "gloo"
