"""bench.py's contract on the GPU box: the JSON line's shape, the N = 1 code path under torch.distributed.run equal to the plain
launch, and the N > 1 path (two ranks sharing the one GPU over gloo -- RCCL needs a GPU per rank) attesting that the metric
all-reduce really spans the ranks, at the worst-case cadence of SURVEY 8(d) config 4 (one collective per env.step())."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "20", "--warmup", "5", "--headline-only"]


def _run(cmd, env=None):
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env={**os.environ, **(env or {})})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_line_shape_and_torchrun_world_1_equals_the_plain_launch():
    plain = _run([sys.executable, "bench.py", "--gpus", "1", *ARGS])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in plain, k
    assert plain["n_gpus"] == 1 and plain["steps"] == 20 and plain["warmup"] == 5 and plain["scaling"] == "weak" and plain["dtype"] == "f32"
    assert plain["vs_baseline"] is None and plain["unit"] == "env-steps/s" and "workload" in plain["config"] and plain["rccl"] is None
    r = plain["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["bound"] == "latency"
    assert abs(plain["value"] - 4096 * 20 / (plain["ms_per_step"] * 20e-3)) < 1e-3 * plain["value"]
    tr = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py", "--gpus", "1", *ARGS])
    assert set(tr) == set(plain) and tr["config"] == plain["config"] and tr["rccl"] is None
    assert tr["episode_metrics"] == plain["episode_metrics"]          # same seeds, same launches: the same episodes end
    assert 0.5 < tr["value"] / plain["value"] < 2.0


def test_two_ranks_reduce_their_metrics_every_step():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29542", "bench.py", "--gpus", "2", *ARGS, "--allreduce-every", "1"], env={"WL_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["total_envs"] == 8192
    rc = line["rccl"]
    assert {k: rc[k] for k in ("backend", "world", "ranks_seen", "allreduce_every")} == {"backend": "gloo", "world": 2, "ranks_seen": 2,
                                                                                      "allreduce_every": 1}
    # what the first multi-GPU run is to yield beside the curve: the per-rank spread of the launch duration and the latency of the
    # path's one collective
    assert 0 < rc["launch_us_min_over_ranks"] <= rc["launch_us_max_over_ranks"] < 100
    ar = rc["metric_allreduce_us"]
    assert ar["bytes"] == 64 and ar["samples"] == 20 and 0 < ar["min_this_rank"] <= ar["median_max_over_ranks"]
    t = line["timing"]
    assert t["allreduce_every"] == 1 and t["metric_reductions_in_timed_blocks"] == 11 * 20      # one collective per env.step()
    assert line["episode_metrics"]["resets"] > 0 and line["roofline"]["envs_per_launch"] == 4096
    for k in ("persistent_rollout", "policy_rollout", "training_iteration"):
        assert line.get(k) is None                    # secondary sections are skipped when world > 1
    assert not line.get("other_tasks")
