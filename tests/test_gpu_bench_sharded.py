"""bench.py's N > 1 path, rehearsed on the one GPU of the test box: two ranks launched the way the driver launches
them (torch.distributed.run, one process per rank), both on device 0 over the shared-memory transport
(SOMAR_BENCH_COMM=shm; RCCL refuses two ranks on one device).  The sharded V-cycle must contract like the one-process
run of the same problem and the JSON line must carry the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_bench_line():
    env = dict(os.environ)
    common = ["--steps", "2", "--warmup", "1", "--size", "128", "--no-cpu-baseline"]  # not "--n": torchrun's parser claims that prefix
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    env2 = dict(env, SOMAR_BENCH_COMM="shm")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2"] + common, env2)
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1
        assert line["metric"] == "pressure-Poisson V-cycles/sec" and line["unit"] == "V-cycles/s"
        assert line["scaling"] == "strong" and line["dtype"] == "f64" and line["vs_baseline"] is None
        assert line["roofline"]["bound"] == "hbm" and 0.0 < line["roofline"]["frac"] < 1.0
        assert line["value"] > 0 and line["ms_per_step"] > 0
        assert line["config"]["cells"] == 128 ** 3
    assert "2 box(es) 128x128x64" in two["config"]["workload"]
    # same problem; the two-box layout stops coarsening one depth earlier (its boxes are 64 cells in z: the reference's
    # coarsenable test works per box), so the cycles differ slightly -- but both contract alike
    assert two["config"]["mg_depth"] == one["config"]["mg_depth"] - 1
    assert two["vcycle_contraction"] == pytest.approx(one["vcycle_contraction"], rel=1e-2)
    # what a sharded run has to reproduce is the SAME two-box layout held by one process (rank 0 computes it in the same run)
    assert one["vcycle_contraction_single_process_same_layout"] is None
    assert two["vcycle_contraction"] == pytest.approx(two["vcycle_contraction_single_process_same_layout"], rel=1e-9)
    assert 0.0 < one["vcycle_contraction"] < 1.0
