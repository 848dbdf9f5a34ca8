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


MAX_LINE = 6144      # bench.MAX_LINE_BYTES: round 4's line grew to 21.9 KB and the driver's record of it had `parsed: null`


def _run(cmd, env=None, detail=None):
    if detail is not None:
        cmd = [*cmd, "--detail-out", str(detail)]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env={**os.environ, **(env or {})})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line, printed by rank 0
    assert out.stdout.strip().splitlines()[-1] == lines[0]          # and it is the LAST line of stdout: what the driver parses
    assert len(lines[0]) < MAX_LINE, len(lines[0])
    return json.loads(lines[0])


def test_the_drivers_own_command_prints_one_short_line_with_roofline_and_cpu_baseline(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5`, every secondary section, sweep and CPU baseline included -- the command
    BENCH_rNN.json records -- ends in ONE `{`-line under 6 KB that carries the contract keys, `roofline` and `cpu_baseline`;
    everything else measured is in the side file the line names"""
    side = tmp_path / "detail.json"
    line = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], detail=side)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_profile", "frac_counters", "wasted_traffic", "valu_frac",
              "kernel", "launch_us", "bytes_per_env_step"):
        assert k in r, k
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["bound"] == "hbm" and r["regime"] == "latency"
    # every fraction on the line follows from numbers on the line: `frac` from the line's own clock (bytes per launch / ms_per_step /
    # peak), `frac_launch` from the event-timed launch, `frac_survey` from SURVEY 8(d)'s 270 B, `frac_profile` from `profile_us`
    # (= one row of the committed kernel summary); the kernel fits in the step
    per_step = r["bytes_per_env_step"] * r["envs_per_launch"] / (line["ms_per_step"] * 1e-3) / 1e9
    assert abs(r["achieved"] - per_step) < 1e-6 * per_step and abs(r["frac"] - per_step / 8000.0) < 1e-9
    assert abs(r["frac_launch"] - r["bytes_per_env_step"] * 4096 / (r["launch_us"] * 1e-6) / 1e9 / 8000.0) < 2e-3 * r["frac_launch"]
    assert r["bytes_per_env_step"] == 334 and r["bytes_per_env_step_survey"] == 270
    assert abs(r["frac_survey"] - r["frac"] * 270 / 334) < 2e-3 * r["frac"]
    if r["frac_profile"] is not None:
        assert abs(r["frac_profile"] - 334 * 4096 / (r["profile_us"] * 1e-6) / 1e9 / 8000.0) < 2e-3 * r["frac_profile"]
    assert r["launch_us"] * 1e-3 <= line["ms_per_step"] * 1.05 and r["frac"] <= r["frac_launch"] * 1.05
    assert abs(line["value"] - 4096 * 20 / (line["ms_per_step"] * 20e-3)) < 1e-6 * line["value"]
    c = line["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "env-steps/s" and c["kind"] == "port" and c["cores"] >= 1 and len(c["sample"]) <= 80
    assert c["full_step_oracle"]["value"] > 0
    assert set(line["other_tasks"]) == {"elevation", "visual", "visual_depth", "visual_depth_task"}
    for t in line["other_tasks"].values():
        assert t["us"] > 0 and 0 < t["frac"] < 1
    assert [s["n"] for s in line["large_n_sweep"]] == [65536, 1048576, 4194304]
    # the side file holds what the line left out (counter dicts, prose, per-section figures)
    d = json.load(open(side))
    assert d["value"] == line["value"] and "timing" in d and "sq_counters" in d["roofline"] and "full_step_oracle" in d["cpu_baseline"]
    assert d["other_tasks"]["elevation"]["roofline"]["kernel"]


def test_line_shape_and_torchrun_world_1_equals_the_plain_launch(tmp_path):
    plain = _run([sys.executable, "bench.py", "--gpus", "1", *ARGS], detail=tmp_path / "a.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in plain, k
    assert plain["n_gpus"] == 1 and plain["steps"] == 20 and plain["warmup"] == 5 and plain["scaling"] == "weak" and plain["dtype"] == "f32"
    assert plain["vs_baseline"] is None and plain["unit"] == "env-steps/s" and "workload" in plain["config"] and plain["rccl"] is None
    r = plain["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["bound"] == "hbm" and r["regime"] == "latency"
    assert abs(plain["value"] - 4096 * 20 / (plain["ms_per_step"] * 20e-3)) < 1e-3 * plain["value"]
    tr = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py", "--gpus", "1", *ARGS], detail=tmp_path / "b.json")
    assert set(tr) == set(plain) and tr["config"] == plain["config"] and tr["rccl"] is None
    assert tr["episode_metrics"] == plain["episode_metrics"]          # same seeds, same launches: the same episodes end
    assert 0.5 < tr["value"] / plain["value"] < 2.0


def test_two_ranks_reduce_their_metrics_every_step(tmp_path):
    short = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", "29542", "bench.py", "--gpus", "2", *ARGS, "--allreduce-every", "1"], env={"WL_BENCH_BACKEND": "gloo"},
                 detail=tmp_path / "c.json")
    assert short["n_gpus"] == 2 and short["config"]["total_envs"] == 8192
    assert short["rccl"]["world"] == 2 and short["rccl"]["ranks_seen"] == 2 and short["rccl"]["metric_allreduce_us"] > 0
    line = json.load(open(tmp_path / "c.json"))          # the full record (rank 0 writes it)
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


@pytest.mark.timeout(900)
def test_eight_ranks_share_the_gpu_through_the_drivers_launcher(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...` -- the command of the driver's 8-GPU lease, which no
    box of this pool has run -- with eight gloo ranks on the one GPU: rank 0 alone prints the one JSON line, the attested world is 8,
    the job is SURVEY 8(d) config 4's 32 768 envs.  (The rate itself says nothing here: eight ranks time-share one device.)"""
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                 "--master-port", "29548", "bench.py", "--gpus", "8", *ARGS], env={"WL_BENCH_BACKEND": "gloo"}, detail=tmp_path / "w8.json")
    assert line["n_gpus"] == 8 and line["config"]["total_envs"] == 32768 and line["config"]["envs_per_gpu"] == 4096
    assert line["rccl"]["world"] == 8 and line["rccl"]["ranks_seen"] == 8 and line["scaling"] == "weak"
    assert abs(line["value"] - 32768 * 20 / (line["ms_per_step"] * 20e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["envs_per_launch"] == 4096          # the roofline stays per GPU: one rank's launch
    assert line["episode_metrics"]["resets"] > 0
