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
