#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused drift env.step() hot path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W           (N>1: launched by torch.distributed.run, one rank / GPU)

A "step" is one env.step() of ALL envs of the shard: one launch of the fused HIP kernel (action term -> 4 physics
sub-steps -> terminations -> rewards -> resets -> pushes -> 14-dim observation).  Workload (config.workload): drift
task, 4096 envs per GPU, flat terrain, synthetic U(-1,1) actions pre-staged in HBM, outputs written into a
[128, n, ...] rollout storage exactly as an on-policy runner would keep them (modified_rsl_rl_runner.py:70-73).
Env shards are independent (weak scaling); the only collective is the episode-metric all-reduce (RCCL) once per
128-step rollout -- the reference's logging cadence (rsl_rl_ppo_cfg.py:6).

The JSON line carries `roofline` (HBM-bound kernel: algorithmic bytes per launch / mean launch duration measured
with HIP events on the launch stream) and `cpu_baseline` (the torch-CPU port of the reference's mdp path, timed on
this box's host cores on a bounded sample; baseline, not target).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ENVS_PER_GPU = 4096
ROLLOUT = 128            # num_steps_per_env of the reference's PPO config
REPEATS = 11             # timed blocks of --steps steps each; the headline is their median
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)

# algorithmic HBM bytes per env-step of the fused drift kernel:
#   reads : 23 dynamic rows + 4 randomisation rows + 7 episode sums (fp32) + episode_len (i32) + action (2 fp32)
#   writes: 23 dynamic rows + 7 episode sums + episode_len + obs 14 fp32 + reward fp32 + terminated u8 + truncated u8
BYTES_READ = (23 + 4 + 7) * 4 + 4 + 8
BYTES_WRITE = (23 + 7) * 4 + 4 + 14 * 4 + 4 + 2
BYTES_PER_ENV_STEP = BYTES_READ + BYTES_WRITE
BYTES_PER_ENV_STEP_SURVEY = 270      # the per-unit figure the survey states for the fused drift step


# algorithmic HBM bytes per unit of the other hot kernels: elevation 270 + 676 x 4
# (height map) + 8 (goal); visual 270 + 3200 x 4 (image); depth ray-cast: the 60 x 80 fp32 image + the camera pose (7 fp32)
ALGO_BYTES = {"drift": BYTES_PER_ENV_STEP, "elev": 2982, "visual": 13070, "depth": 60 * 80 * 4 + 28,
              "visual_depth": 270 + (60 * 80 + 8) * 4}     # the visual-depth task step: state rows + the 4808-float observation row


_SHARED_SRC = ["wl_kernel_common.h", "wl_math.h", "wl_rng.h", "wl_vehicle.h", "wl_drift_terms.h"]
TASK_SOURCES = {"drift": ["wl_drift.hip", "wl_drift_env.h"] + _SHARED_SRC,
                "elev": ["wl_elev.hip", "wl_heightfield.h", "wl_actor_dev.h", "wl_mlp.h"] + _SHARED_SRC,
                "visual": ["wl_visual.hip"] + _SHARED_SRC,
                "depth": ["wl_depth.hip", "wl_depth_dev.h", "wl_heightfield.h", "wl_kernel_common.h", "wl_math.h"]}
TASK_SOURCES["visual_depth"] = sorted(set(TASK_SOURCES["visual"] + TASK_SOURCES["depth"]))


def csrc_fingerprint(task: str):
    """sha256 over the sources a task's kernels are built from (+ the build flags): stamps the rocprofv3 --pmc digests under
    profiles/ so that counters measured on an older build of THAT kernel are marked stale instead of riding silently next to
    live timings"""
    import hashlib
    h = hashlib.sha256()
    for f in TASK_SOURCES[task] + ["../../include/wheeledlab_amd.h", "../../__graft_entry__.py"]:
        path = os.path.normpath(os.path.join(ROOT, "wheeledlab_amd", "csrc", f))
        data = open(path, "rb").read()
        if f.endswith(".py"):   # only the compiler flags matter
            data = b"\n".join(l for l in data.splitlines() if b"HIPCC_FLAGS" in l or b"-mllvm" in l)
        else:                   # code only: comments and blank space do not change a kernel
            import re
            data = re.sub(rb"/\*.*?\*/", b"", data, flags=re.S)
            data = re.sub(rb"//[^\n]*", b"", data)
            data = b"\n".join(l.strip() for l in data.splitlines() if l.strip())
        h.update(f.encode())
        h.update(data)
    return h.hexdigest()[:16]


def pmc_entry(task: str, n_envs: int):
    """the newest committed counter digest (profiles/r*_pmc.json, tools/pmc_report.py) for (task, env count), or None;
    `stale` when the kernels have changed since it was measured"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        d = json.load(open(f))
        e = d.get("entries", {}).get(f"{task}:{n_envs}")
        if e:
            e = dict(e)
            e["source"] = os.path.relpath(f, ROOT)
            e["stale"] = d.get("csrc_fingerprints", {}).get(task) != csrc_fingerprint(task)
            return e
    return None


def rocprof_avg_us(kernels):
    """average duration of a kernel (or the sum over a list of kernels: the launches of one step) in the newest committed
    `rocprofv3 --kernel-trace --stats` summary of bench.py itself (profiles/r*_bench_kernel_stats.csv), for the reader who
    recomputes `frac` from profiles/.  A kernel is named by the substrings its row's Name must ALL contain -- a string, or a
    tuple such as ("visual_step_kernel<", "FlatGround>") that pins the exact instantiation a section timed (the flat-ground
    visual step, not the heightfield one of the depth task).  A tuple -- of any length, one element included -- that still matches
    several rows is ambiguous: None."""
    import csv
    import glob
    names = [kernels] if isinstance(kernels, (str, tuple)) else list(kernels)
    bare = [isinstance(k, str) for k in names]       # only a BARE string may fall back to its most-called instantiation
    names = [(k,) if isinstance(k, str) else tuple(k) for k in names]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")), reverse=True):
        try:
            rows = list(csv.DictReader(open(f)))
            tot, calls = 0.0, None
            for subs, is_bare in zip(names, bare):
                hits = [r for r in rows if all(sub in r.get("Name", "") for sub in subs)]
                if len(hits) > 1 and is_bare:      # a bare name: the most-called instantiation (the headline's 10^4 launches)
                    hits = sorted(hits, key=lambda r: -int(r["Calls"]))[:1]
                if len(hits) != 1:
                    tot = None
                    break
                tot += float(hits[0]["AverageNs"]) / 1e3
                c = int(hits[0]["Calls"])
                calls = c if calls is None else min(calls, c)
            if tot is not None:
                return {"avg_us": tot, "calls": calls, "source": os.path.relpath(f, ROOT)}
        except (KeyError, ValueError):
            continue
    return None


def roofline_block(task, n, us, kernel, bound, bytes_per_unit=None, profile_kernels=None):
    """HBM roofline of one kernel / step: algorithmic bytes per launch / live duration, + three figures a reader can check
    without opening a CSV: `frac_profile` (the same bytes / the kernel's average duration in the committed rocprofv3 summary of
    bench.py, profiles/r*_bench_kernel_stats.csv), `frac_counters` (HBM bytes the PMC passes counted / the kernel's duration in
    those passes / peak) and `wasted_traffic` (counter bytes / algorithmic bytes: re-reads and write-allocate overhead)"""
    bpu = bytes_per_unit or ALGO_BYTES[task]
    achieved = bpu * n / (us * 1e-6) / 1e9
    e = pmc_entry(task, n)
    blk = {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": None, "kernel": kernel, "launch_us": us, "bytes_per_unit": bpu, "units_per_launch": n,
           "frac_uses": "launch_us (HIP events on the launch stream, live in this run)",
           "frac_profile": None, "frac_counters": None, "wasted_traffic": None}
    if profile_kernels:
        st = rocprof_avg_us(profile_kernels)
        if st:
            blk["rocprof_kernel_stats"] = st
            blk["frac_profile"] = bpu * n / (st["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    if e:
        blk["traffic"] = None if e["stale"] else e.get("traffic_bytes")
        blk["traffic_source"], blk["traffic_stale"] = e["source"], e["stale"]
        if not e["stale"]:
            blk["counters"] = e.get("kernels")
            blk["frac_counters"] = e.get("frac_counters")
            if e.get("traffic_bytes") and bytes_per_unit is None:
                blk["wasted_traffic"] = e["traffic_bytes"] / (bpu * n)
    return blk


def pmc_traffic(n_envs: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json); bench.py cannot
    profile itself, so the number is the one measured with tools/pmc_run.py on the same kernel and env count."""
    e = pmc_entry("drift", n_envs)
    if e and e.get("traffic_bytes"):
        return (None if e["stale"] else e["traffic_bytes"]), e["source"], e["stale"]
    return None, None, None


def pmc_sq(n_envs: int):
    """SQ-counter view of the same kernel and env count from the committed rocprofv3 --pmc passes
    (profiles/r*_pmc.json, made by tools/profile_pmc.sh): VALU instructions per wavefront, the fractions of a
    wavefront's life spent issuing VALU / parked in s_waitcnt / stalled at issue, and the share of the VALU pipe's time
    the instruction mix occupies -- the second roofline of this kernel (it is not bandwidth-shaped at 4096 envs)."""
    e = pmc_entry("drift", n_envs)
    if e and not e["stale"]:
        k = dict(e.get("kernels", {}).get("drift_step_kernel", {}))
        if k:
            k["source"] = e["source"]
            return k
    return None


def cpu_baseline(n_envs: int, budget_s: float = 14.0):
    """The baseline BASELINE.json's north_star and SURVEY 8(d) name: the torch-CPU port of the reference's drift mdp path
    (oracle/torch_mdp.py -- action term, terminations, rewards, noisy observation: what the reference itself computes
    outside PhysX) on the GPU box's host cores.  Beside it (`full_step_oracle`) the numpy oracle's FULL env.step (the same
    workload as the fused kernel: + 4 integrator sub-steps of the vehicle model + in-step reset) on one thread."""
    host_cores = os.cpu_count() or 1
    out = _cpu_mdp_only(n_envs, budget_s * 0.5, host_cores)
    out["full_step_oracle"] = _cpu_full_step(n_envs, budget_s * 0.5, host_cores)
    return out


def _cpu_full_step(n_envs: int, budget_s: float, host_cores: int):
    import numpy as np

    from oracle import drift_reset as DR
    from oracle import drift_step as OS
    from oracle import params as OP

    torch.set_num_threads(1)
    p = OP.drift_params()
    rng = np.random.RandomState(0)
    st = OS.init_state(p, n_envs, seed=0)
    ep = np.zeros(st.shape[1], np.int32)
    ref = DR.ref_pose_table(DR.reference_poses(rng.rand(int(p.num_ref_points)).astype(np.float32)))
    OS.reset_envs(p, st, ep, ref, np.arange(n_envs), 42, 0)
    acts = (rng.rand(8, n_envs, 2) * 2 - 1).astype(np.float32)
    OS.step(p, st, ep, ref, acts[0], 42, 0)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s:
        OS.step(p, st, ep, ref, acts[k % 8], 42, k + 1)
        k += 1
    dt = time.perf_counter() - t0
    return {"value": n_envs * k / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{k} full drift env.steps (4 sub-steps of the vehicle model + mdp terms + in-step reset + noisy "
                      f"obs) of the numpy oracle on {n_envs} envs, numpy {np.__version__}, 1 thread of a {host_cores}-core "
                      f"host, {dt:.1f} s"}


def _cpu_mdp_only(n_envs: int, budget_s: float, host_cores: int):
    """torch-CPU port of the reference's drift mdp path (oracle/torch_mdp.py) on the host cores."""
    from oracle import torch_mdp as T

    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.rand(*s, generator=g)
    pos = torch.cat([r(n_envs, 2) * 4 - 2, torch.zeros(n_envs, 1)], -1)
    yaw = r(n_envs) * 6.28
    quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n_envs), torch.zeros(n_envs), torch.sin(yaw / 2)], -1)
    vb, wb, ww = r(n_envs, 3) * 3, r(n_envs, 3) - 0.5, r(n_envs, 3) - 0.5
    steer = r(n_envs, 2) - 0.5
    act = r(n_envs, 2) * 2 - 1
    ep = torch.zeros(n_envs, dtype=torch.int32)
    w = [10.0, -5.0, 40.0, 0.0, 20.0, -50.0, -5000.0]
    scale, off = torch.tensor([3.0, 0.488]), torch.zeros(2)
    best = None
    per_threads = {}
    # k in {1, all cores} (+ a modest pool): small elementwise ops do not scale with intra-op threads, so
    # every setting is reported and the fastest is the headline baseline
    settings = sorted({1, min(8, host_cores), host_cores})
    for cores in settings:
        torch.set_num_threads(cores)
        with torch.inference_mode():
            for _ in range(5):
                T.mdp_step(pos, quat, vb, wb, ww, steer, act, ep, w, scale, off)
            t0 = time.perf_counter()
            iters = 0
            while time.perf_counter() - t0 < budget_s / len(settings):
                for _ in range(10):
                    T.mdp_step(pos, quat, vb, wb, ww, steer, act, ep, w, scale, off)
                iters += 10
            dt = time.perf_counter() - t0
        per_threads[str(cores)] = n_envs * iters / dt
        if best is None or iters / dt > best[1] / best[2]:
            best = (cores, iters, dt)
    cores, iters, dt = best
    return {"value": n_envs * iters / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "host_cores": host_cores, "cpu_model": _cpu_model(), "by_threads": per_threads,
            "sample": f"{iters} passes of the drift mdp path only (action term + 2 terminations + 7 rewards + noisy 14-dim "
                      f"obs; no physics) on {n_envs} envs, torch {torch.__version__} CPU, best of "
                      f"{' / '.join(str(c) for c in settings)} intra-op threads, {dt:.1f} s"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


SWEEP_TESTS = {65536: "tests/test_gpu_forms.py::test_drift_65536_envs_equal_two_shards",
               1048576: "tests/test_gpu_forms.py::test_drift_4m_envs_equal_four_1m_shards_and_the_oracle (the shards)",
               4194304: "tests/test_gpu_forms.py::test_drift_4m_envs_equal_four_1m_shards_and_the_oracle + "
                        "test_streaming_drift_form_matches_oracle_single_steps"}


def large_n_sweep(dev):
    """us per fused drift env.step() at 65 536 / 1 M / 4 M envs (outputs overwritten in place; no int64 `dones` row: the 334 B
    per env-step of the headline kernel).  `test`: the GPU test that runs the same size and kernel instantiation."""
    from wheeledlab_amd.core import DriftBatch

    sweep = []
    for big in (65536, 1048576, 4194304):
        e2 = DriftBatch(big, device=dev, seed=42)
        e2.set_dones_output(False)
        e2.reset()
        a2 = torch.rand(8, big, 2, device=dev) * 2 - 1
        for _ in range(12):     # ~ 30 ms of launches at 4 M envs: clocks and the caches' contents settle (3 were not enough:
            e2.rollout(a2)      # the same kernel measured 284 us under rocprofv3's counters and 292 - 315 us here)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(4):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(6):
                e2.rollout(a2)
            s1.record()
            torch.cuda.synchronize()
            best = min(best, s0.elapsed_time(s1) * 1e3 / 48)
        gbs = BYTES_PER_ENV_STEP * big / (best * 1e-6) / 1e9
        sq = pmc_sq(big)
        pe = pmc_entry("drift", big)
        fresh = pe is not None and not pe["stale"]
        sweep.append({"n_envs": big, "us_per_step": round(best, 2), "env_steps_per_s": big / (best * 1e-6),
                      "achieved_GBs": gbs, "frac_of_8TBs": gbs / 8000.0, "bytes_per_env_step": BYTES_PER_ENV_STEP,
                      "frac_counters": pe.get("frac_counters") if fresh else None,
                      "wasted_traffic": pe["traffic_bytes"] / (BYTES_PER_ENV_STEP * big) if fresh and pe.get("traffic_bytes") else None,
                      "counters_source": pe["source"] if pe else None, "counters_stale": pe["stale"] if pe else None,
                      "bound": "valu+hbm" if big >= 1048576 else "latency+valu",
                      "valu_frac": sq.get("valu_pipe_frac") if sq else None, "sq_counters": sq, "test": SWEEP_TESTS[big]})
        del e2, a2
    return sweep


OTHER_SWEEP_TESTS = {("elev", 262144): "tests/test_gpu_forms.py::test_elevation_262144_envs_equal_two_shards + "
                                        "test_height_scan_forms_are_bit_identical",
                     ("visual", 65536): "tests/test_gpu_forms.py::test_visual_65536_envs_equal_two_shards + "
                                        "test_streaming_camera_rows_are_bit_identical"}


def other_tasks_sweep(dev):
    """the elevation / visual steps and the depth ray-cast at env counts past the latency regime (per-step launches, outputs
    overwritten in place): the HBM fraction of SURVEY 8(d)'s bytes per unit"""
    from wheeledlab_amd.core import DepthCamera, ElevBatch, VisualBatch

    out = []
    for task, cls, sizes, K in (("elev", ElevBatch, (65536, 262144, 1048576), 4), ("visual", VisualBatch, (65536, 262144), 2)):
        for big in sizes:
            e2 = cls(big, device=dev, seed=42)
            e2.reset()
            if task == "visual":
                e2.sample_augmentation(torch.Generator().manual_seed(0))
            a2 = torch.rand(K, big, 2, device=dev) * 2 - 1
            for _ in range(4):
                e2.rollout(a2)
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(3):
                    e2.rollout(a2)
                s1.record()
                torch.cuda.synchronize()
                best = min(best, s0.elapsed_time(s1) * 1e3 / (3 * K))
            blk = roofline_block(task, big, best, "step + scan" if task == "elev" else "step + camera", "hbm+valu")
            out.append({"task": task, "n_envs": big, "us_per_step": round(best, 2), "env_steps_per_s": big / (best * 1e-6),
                        "achieved_GBs": blk["achieved"], "frac_of_8TBs": blk["frac"], "bytes_per_env_step": ALGO_BYTES[task],
                        "traffic": blk.get("traffic"), "traffic_stale": blk.get("traffic_stale"),
                        "frac_counters": blk.get("frac_counters"), "wasted_traffic": blk.get("wasted_traffic"),
                        "test": OTHER_SWEEP_TESTS.get((task, big))})
            if task == "elev" and big == 65536:
                cam = DepthCamera(e2.hf, dev)
                img = torch.empty(big, 60, 80, device=dev)
                cam.render(e2, 100.0, img)
                torch.cuda.synchronize()
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(3):
                    cam.render(e2, 100.0, img)
                s1.record()
                torch.cuda.synchronize()
                dus = s0.elapsed_time(s1) * 1e3 / 3
                blk = roofline_block("depth", big, dus, "visual_depth_tile_kernel", "valu+latency")
                out.append({"task": "depth", "n_envs": big, "us_per_render": round(dus, 2), "rays_per_s": big * 4800 / (dus * 1e-6),
                            "achieved_GBs": blk["achieved"], "frac_of_8TBs": blk["frac"], "bytes_per_image": ALGO_BYTES["depth"]})
                del cam, img
            del e2, a2
            torch.cuda.empty_cache()
    return out


MAX_LINE_BYTES = 6144     # the driver's parser lost round 4's 21.9 KB line: the final stdout line stays under this


def _r(x, sig=4):
    """floats to `sig` significant digits (the compact line; the side file keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d.get(k) for k in keys} if d else None


def write_detail(detail, path=None):
    """everything bench.py measured, with the per-kernel counter dicts and the prose, as a side file (default
    gpurun_out/bench_detail.json: gpurun merges that directory back); returns the path relative to the repo, or None"""
    path = path or os.environ.get("WL_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as ex:     # a read-only checkout must not cost the line
        print(f"[bench] could not write {path}: {ex!r}", file=sys.stderr, flush=True)
        return None


_ROOF_KEYS = ("bound", "regime", "achieved", "peak", "unit", "frac", "traffic", "frac_launch", "frac_profile", "frac_counters", "frac_survey",
              "wasted_traffic", "valu_frac", "kernel", "launch_us", "profile_us", "bytes_per_env_step", "bytes_per_env_step_survey",
              "envs_per_launch", "frac_is")
_TASK_ROOF_KEYS = ("frac", "frac_profile", "frac_counters", "wasted_traffic", "valu_frac")


def _task_valu_frac(roof):
    """VALU-pipe share of the task's dominant kernel from the committed counter digest (the largest over the step's kernels)"""
    if not roof:
        return None
    if roof.get("valu_frac") is not None:
        return roof["valu_frac"]
    v = [k.get("valu_pipe_frac") for k in (roof.get("counters") or {}).values() if k.get("valu_pipe_frac") is not None]
    return max(v) if v else None


def compact_line(d, detail_path):
    """the ONE JSON line the driver parses: contract keys + config + roofline + cpu_baseline + per-task / per-size digests;
    no prose beyond the workload name, no counter dicts (those are in the side file `detail`)"""
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data")}
    c = d["config"]
    line["config"] = {"workload": f"drift task, {c['envs_per_gpu']} envs/GPU, flat terrain, fused step kernel, synthetic actions in HBM",
                      "envs_per_gpu": c["envs_per_gpu"], "total_envs": c["total_envs"], "parallelism": f"env-shard x{d['n_gpus']}"}
    line["roofline"] = _pick(d["roofline"], _ROOF_KEYS)
    cb = d.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind", "host_cores")),
                                "sample": f"{cb['sample'].split(' ')[0]} drift mdp passes (no physics), {c['envs_per_gpu']} envs, torch-CPU"[:80],
                                "full_step_oracle": _pick(cb.get("full_step_oracle"), ("value", "cores"))}
    if d.get("rccl"):
        rc = d["rccl"]
        line["rccl"] = {**_pick(rc, ("backend", "world", "ranks_seen", "allreduce_every", "launch_us_min_over_ranks", "launch_us_max_over_ranks")),
                        "metric_allreduce_us": (rc.get("metric_allreduce_us") or {}).get("median_max_over_ranks")}
    else:
        line["rccl"] = None
    line["episode_metrics"] = d["episode_metrics"]
    line["gpu_event_ms_per_step"] = d["gpu_event_ms_per_step"]
    ot = {}
    for name, o in (d.get("other_tasks") or {}).items():
        roof = o.get("roofline") or {}
        e = {"us": o.get("us_per_step", o.get("us_per_render")), **_pick(roof, _TASK_ROOF_KEYS)}
        e["valu_frac"] = _task_valu_frac(roof)
        ot[name] = e
    if ot:
        line["other_tasks"] = ot
    if d.get("large_n_sweep"):
        line["large_n_sweep"] = [{"n": r["n_envs"], "us": r["us_per_step"], "frac": r["frac_of_8TBs"], "frac_counters": r.get("frac_counters"),
                                  "valu_frac": r.get("valu_frac"), "test": r["test"].split("::")[-1].split(" ")[0]} for r in d["large_n_sweep"]]
    if d.get("other_tasks_large_n"):
        line["other_tasks_large_n"] = [{"task": r["task"], "n": r["n_envs"], "us": r.get("us_per_step", r.get("us_per_render")),
                                        "frac": r["frac_of_8TBs"], "frac_counters": r.get("frac_counters")} for r in d["other_tasks_large_n"]]
    for k, sub in (("persistent_rollout", "env_steps_per_s"), ("policy_rollout", "env_steps_per_s"), ("training_iteration", "env_steps_per_s")):
        if d.get(k):
            line[k + "_env_steps_per_s"] = d[k][sub]
    if d.get("python_surface_env_steps_per_s"):
        line["python_surface_env_steps_per_s"] = d["python_surface_env_steps_per_s"]
    line["detail"] = detail_path
    out = _r(line)
    # the contract's own figures keep full precision: the driver recomputes value from ms_per_step
    for k in ("value", "ms_per_step"):
        out[k] = d[k]
    out["roofline"]["achieved"], out["roofline"]["frac"] = d["roofline"]["achieved"], d["roofline"]["frac"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allreduce-every", type=int, default=ROLLOUT,
                    help="steps between episode-metric all-reduces (default 128 = the reference's logging cadence; 1 = SURVEY 8(d) "
                         "config 4's worst case: one collective per env.step())")
    ap.add_argument("--sweep", action="store_true", help="(default at N=1) large-N sweep of the same kernel: the HBM-bound regime")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--sweep-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline workload: no secondary sections, no CPU baseline")
    ap.add_argument("--detail-out", default=None, help="side file for everything measured (default gpurun_out/bench_detail.json)")
    ap.add_argument("--print-detail", action="store_true", help="also print the full detail as an earlier stdout line prefixed `[bench detail] `")
    args = ap.parse_args()
    if args.sweep_child:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
        d0 = torch.device("cuda", 0)
        print(json.dumps({"drift": large_n_sweep(d0), "other": other_tasks_sweep(d0)}), flush=True)
        return

    if args.headline_only:
        args.no_sweep = args.no_cpu_baseline = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # WL_BENCH_BACKEND=gloo is a debugging aid only (lets the N>1 code path run with several ranks sharing one GPU)
    backend = os.environ.get("WL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from wheeledlab_amd.core import DriftBatch

    n = args.envs_per_gpu
    env = DriftBatch(n, device=dev, seed=42, env_offset=rank * n)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    actions = torch.rand(ROLLOUT, n, 2, device=dev, generator=g) * 2 - 1
    obs_buf = torch.zeros(ROLLOUT, n, 14, device=dev)
    rew_buf = torch.zeros(ROLLOUT, n, device=dev)
    term_buf = torch.zeros(ROLLOUT, n, dtype=torch.uint8, device=dev)
    trunc_buf = torch.zeros(ROLLOUT, n, dtype=torch.uint8, device=dev)
    metric_sum = torch.zeros_like(env.metrics)

    # The episode-metric all-reduce (the path's only collective) runs at the LOGGING cadence -- once per ROLLOUT = 128 steps,
    # the reference's num_steps_per_env -- whatever --steps is: a 20-step timed block does not contain 1 / 20 th of a
    # logging event, it contains one every 6.4 blocks.  It is issued async on RCCL's own stream and joined one logging
    # interval later, so the next 128 launches overlap it instead of queueing behind it.  drain() (after the timing)
    # reduces what is left so that short runs report their episode metrics too.
    every = max(1, min(args.allreduce_every, ROLLOUT))
    cadence = {"since": 0, "pending": None, "reductions": 0}

    def join():
        p = cadence["pending"]
        if p is not None:
            p[0].wait()
            metric_sum.add_(p[1])
            cadence["pending"] = None

    def reduce_metrics():
        m = env.read_metrics(zero=True)
        if dist is not None:
            join()
            cadence["pending"] = (dist.all_reduce(m, async_op=True), m)
        else:
            metric_sum.add_(m)
        cadence["since"] = 0
        cadence["reductions"] += 1

    def run(k_steps):
        done = 0
        while done < k_steps:
            k = min(every - cadence["since"], k_steps - done)
            env.rollout(actions[:k], obs_buf, rew_buf, term_buf, trunc_buf)
            done += k
            cadence["since"] += k
            if cadence["since"] >= every:
                reduce_metrics()

    def drain():
        if cadence["since"] > 0:
            reduce_metrics()
        join()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # attestation that the collective really spans `world` ranks on the RCCL backend: a sum all-reduce of ones
    rccl = None
    if dist is not None:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_seen": int(ones.item()),
                "allreduce_every": every}
        if rccl["ranks_seen"] != world or rccl["world"] != world:
            # a collective that does not span the job would make `value` a lie: fail the run instead of printing a line
            print(f"[bench] rank {rank}: the all-reduce saw {rccl['ranks_seen']} ranks in a world of {rccl['world']}, launched as {world}",
                  file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)
    run(args.warmup)
    # EXACTLY --steps steps per timed block, bracketed by barrier + synchronize, max over ranks -- and REPEATS such blocks:
    # at the driver's K = 20 one block is 0.15 ms, a single sample of which is launch-queue noise (round 1 reported
    # 10.9 us / step where the steady state was 7.2); the headline is the median block
    walls, gpu = [], []
    cadence["reductions"] = 0
    for _ in range(REPEATS):
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        w = time.perf_counter() - t0
        t = torch.tensor([w], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls.append(float(t.item()))
    reductions_in_timed_blocks = cadence["reductions"]
    # the same block between two events on the launch stream (secondary figure; NOT inside the wall-timed blocks above:
    # two event records are 8 us of host work, 6 % of a 20-step block)
    for _ in range(5):
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        run(args.steps)
        ev1.record()
        barrier()
        gpu.append(ev0.elapsed_time(ev1))
    wall = sorted(walls)[len(walls) // 2]
    wall_mean = sum(walls) / len(walls)
    gpu_ms = sorted(gpu)[len(gpu) // 2]
    drain()   # what the timed blocks left in the accumulators (outside the timing: short runs report their metrics too)

    # kernel-only duration: K back-to-back launches of the fused step between two events on the launch stream
    env.rollout(actions, obs_buf, rew_buf, term_buf, trunc_buf)
    torch.cuda.synchronize()
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 8
    k0.record()
    for _ in range(reps):
        env.rollout(actions, obs_buf, rew_buf, term_buf, trunc_buf)
    k1.record()
    torch.cuda.synchronize()
    launch_us = k0.elapsed_time(k1) * 1e3 / (reps * ROLLOUT)
    # N > 1: the spread of the per-rank launch duration (a slow GPU shows here, not in the max-over-ranks headline) and the
    # event-timed latency of ONE episode-metric all-reduce on an idle stream -- the path's only collective, so that the first
    # multi-GPU run yields the scaling curve and the collective's cost in one shot
    if dist is not None:
        lu = torch.tensor([launch_us, -launch_us], device=dev, dtype=torch.float64)
        dist.all_reduce(lu, op=dist.ReduceOp.MAX)
        m = env.read_metrics(zero=False)
        for _ in range(3):
            dist.all_reduce(m)
        barrier()
        lat = []
        for _ in range(20):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            dist.all_reduce(m)
            a1.record()
            torch.cuda.synchronize()
            lat.append(a0.elapsed_time(a1) * 1e3)
        lat.sort()
        lt = torch.tensor([lat[len(lat) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(lt, op=dist.ReduceOp.MAX)
        rccl.update({"launch_us_max_over_ranks": float(lu[0]), "launch_us_min_over_ranks": float(-lu[1]),
                     "metric_allreduce_us": {"median_max_over_ranks": float(lt.item()), "min_this_rank": lat[0], "max_this_rank": lat[-1],
                                             "bytes": int(m.numel() * m.element_size()), "samples": len(lat),
                                             "timing": "HIP events around one blocking all_reduce of the [16] metric vector on an idle stream"}})
    # The line's own fraction follows from the line's own clock: algorithmic bytes of one launch / ms_per_step (the wall time per
    # step of the timed blocks, barrier + synchronise bracket included) / peak.  The same bytes over the kernel's duration between
    # two events on the launch stream are `frac_launch`; over its average in the committed rocprofv3 summary `frac_profile`.
    step_us = wall * 1e6 / args.steps
    achieved = BYTES_PER_ENV_STEP * n / (step_us * 1e-6) / 1e9
    achieved_launch = BYTES_PER_ENV_STEP * n / (launch_us * 1e-6) / 1e9
    traffic, traffic_src, traffic_stale = pmc_traffic(n)
    sq = pmc_sq(n)
    # which roofline binds: at the BASELINE size the state is L2 / Infinity-Cache resident and one wavefront per SIMD
    # issues dependent instructions -- launch floor + instruction latency, not bandwidth; the HBM fraction is reported
    # regardless (the contract's metric) and the sweep below shows the regime where bandwidth and the VALU pipe bind
    regime = "latency" if n <= 32768 else "valu+hbm"

    # secondary: the same K-step rollouts as ONE persistent launch each (state in registers across steps; only possible
    # with pre-staged actions, so it is NOT the headline; the policy-in-the-loop form follows)
    persistent = None
    if rank == 0 and world == 1 and n <= 32768 and not args.headline_only:
        env.rollout(actions, obs_buf, rew_buf, term_buf, trunc_buf, persistent=True)
        torch.cuda.synchronize()
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record()
        for _ in range(reps):
            env.rollout(actions, obs_buf, rew_buf, term_buf, trunc_buf, persistent=True)
        p1.record()
        torch.cuda.synchronize()
        pus = p0.elapsed_time(p1) * 1e3 / (reps * ROLLOUT)
        persistent = {"us_per_step": pus, "env_steps_per_s": n / (pus * 1e-6), "steps_per_launch": ROLLOUT,
                      "kernel": "drift_rollout_kernel<FlatGround>"}

    # secondary: the reference runner's whole collection loop (actor MLP -> sample -> env.step, 128 steps per env,
    # rsl_rl_ppo_cfg.py:6) as ONE launch with the actor on the f32 matrix pipe + the critic over the stored observations;
    # next to it the same loop with per-step launches and the actor in torch (what a rsl_rl user runs today)
    policy = None
    if rank == 0 and world == 1 and n <= 32768 and not args.headline_only:
        from wheeledlab_amd.policy import ActorCritic, RolloutStorage
        ac = ActorCritic(device=dev, seed=0)
        store = RolloutStorage(ROLLOUT, n, device=dev)
        env.observe()
        env.rollout_policy(ac, store)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(reps):
            env.rollout_policy(ac, store)
        c1.record()
        torch.cuda.synchronize()
        cus = c0.elapsed_time(c1) * 1e3 / (reps * ROLLOUT)
        lin = [torch.nn.Linear(14, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 2)]
        actor_t = torch.nn.Sequential(lin[0], torch.nn.ELU(), lin[1], torch.nn.ELU(), lin[2]).to(dev)
        obs_t = env.obs
        with torch.inference_mode():
            for timed in (False, True):
                if timed:
                    torch.cuda.synchronize()
                    c0.record()
                for _ in range(ROLLOUT):
                    mu = actor_t(obs_t)
                    a_t = mu + ac.std * torch.randn_like(mu)
                    obs_t, _, _, _ = env.step(a_t)
            c1.record()
            torch.cuda.synchronize()
        tus = c0.elapsed_time(c1) * 1e3 / ROLLOUT
        policy = {"us_per_step": cus, "env_steps_per_s": n / (cus * 1e-6), "steps_per_launch": ROLLOUT,
                  "includes": "actor 14-64-64-2 ELU (fp32 MFMA), Gaussian sampling, log-prob, env.step, storage rows, "
                              "critic over K+1 observations",
                  "kernel": "drift_policy_rollout_kernel<ELU, FlatGround> + mlp_forward_kernel<ELU>",
                  "per_step_launch_torch_actor_us_per_step": tus,
                  "per_step_launch_torch_actor_env_steps_per_s": n / (tus * 1e-6)}

    # secondary: whole training iterations (fused collection + GAE + 20 fused PPO minibatch steps), env-steps/s end to end
    train = None
    if rank == 0 and world == 1 and n <= 32768 and not args.headline_only:
        import wheeledlab_amd.tasks  # noqa: F401
        from wheeledlab_amd import registry
        from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
        from wheeledlab_amd.rl.ppo import OnPolicyRunner
        tcfg = registry.parse_env_cfg("Isaac-MushrDriftRL-v0", device=str(dev), num_envs=n)
        te = registry.make("Isaac-MushrDriftRL-v0", cfg=tcfg)
        te.action_space.low, te.action_space.high = -1.0, 1.0
        runner = OnPolicyRunner(RslRlVecEnvWrapper(ClipAction(te)),
                                registry.load_cfg_from_registry("Isaac-MushrDriftRL-v0", "rsl_rl_cfg_entry_point"), device=str(dev))
        runner.learn(3, verbose=False)
        torch.cuda.synchronize()
        tt = time.perf_counter()
        hist = runner.learn(12, verbose=False)
        torch.cuda.synchronize()
        tt = time.perf_counter() - tt
        train = {"env_steps_per_s": 12 * runner.num_steps_per_env * n / tt, "ms_per_iteration": tt / 12 * 1e3,
                 "collection_ms": 1e3 * sum(h["collection_time"] for h in hist[-12:]) / 12,
                 "learn_ms": 1e3 * sum(h["learn_time"] for h in hist[-12:]) / 12, "fused_collection": runner.fused,
                 "fused_learner": runner.alg.fused_update, "steps_per_env": runner.num_steps_per_env}
        del runner, te

    # secondary: the other two tasks at the same env count (configs[2] and [4] of BASELINE.json), per-step launches
    other = {}
    if rank == 0 and world == 1 and not args.headline_only:
        from wheeledlab_amd.core import ElevBatch, VisualBatch
        for name, cls, k in (("elevation", ElevBatch, 32), ("visual", VisualBatch, 16)):
            t = cls(n, device=dev, seed=42)
            t.reset()
            if name == "visual":
                t.sample_augmentation(torch.Generator().manual_seed(0))   # the reference's default obs term is the augmented one
            a = torch.rand(k, n, 2, device=dev) * 2 - 1
            t.rollout(a)
            torch.cuda.synchronize()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            us = 1e30
            for _ in range(3):      # best of three blocks: one block once measured 12 x the others (a host hiccup in a 3 ms window)
                q0.record()
                for _ in range(4):
                    t.rollout(a)
                q1.record()
                torch.cuda.synchronize()
                us = min(us, q0.elapsed_time(q1) * 1e3 / (4 * k))
            other[name] = {"us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "obs_dim": t.OBS_DIM,
                           "obs_GBs": n * t.OBS_DIM * 4 / (us * 1e-6) / 1e9, "launches": "one per env.step()" if name == "elevation"
                           else "two per env.step() (step, camera)"}
            # roofline of the whole step (SURVEY 8(d) bytes per env-step / live us per step) and, for the visual task, of its
            # dominant kernel alone (the camera: 12 832 B of observation row per env)
            if name == "elevation":
                other[name]["roofline"] = roofline_block("elev", n, us, "elev_step_scan_kernel" if n <= 32768 else
                                                         "elev_step_kernel + elev_scan_kernel",
                                                         "latency (10 dependent integrator sub-steps) + texture-unit address rate (height scan)" if n <= 32768 else "hbm+valu+texture-unit address rate",
                                                         profile_kernels=["elev_step_scan_kernel"] if n == ENVS_PER_GPU else None)
            else:
                other[name]["roofline"] = roofline_block("visual", n, us, "visual_step_kernel + visual_obs_kernel", "hbm+lds",
                                                         profile_kernels=[("visual_step_kernel<", "FlatGround>("), "visual_obs_kernel"] if n == ENVS_PER_GPU else None)
                t.observe()
                torch.cuda.synchronize()
                q0.record()
                for _ in range(32):
                    t.observe()
                q1.record()
                torch.cuda.synchronize()
                cam_us = q0.elapsed_time(q1) * 1e3 / 32
                other[name]["camera_roofline"] = roofline_block("visual", n, cam_us, "visual_obs_kernel", "hbm+lds",
                                                                bytes_per_unit=t.OBS_DIM * 4 + 60)
            if n <= 32768:
                # open-loop rollouts (pre-staged actions) as ONE launch: the height scan / camera of step k while step k + 1 is
                # integrated (wl_elev_rollout_persistent, wl_visual_rollout_persistent; same results bit for bit)
                po = torch.zeros(k, n, t.OBS_DIM, device=dev)
                pr = torch.zeros(k, n, device=dev)
                pt, pu = torch.zeros(k, n, dtype=torch.bool, device=dev), torch.zeros(k, n, dtype=torch.bool, device=dev)
                t.rollout(a, po, pr, pt, pu, persistent=True)
                torch.cuda.synchronize()
                q0.record()
                for _ in range(4):
                    t.rollout(a, po, pr, pt, pu, persistent=True)
                q1.record()
                torch.cuda.synchronize()
                pus = q0.elapsed_time(q1) * 1e3 / (4 * k)
                other[name]["persistent_rollout_us_per_step"] = pus
                other[name]["persistent_rollout_env_steps_per_s"] = n / (pus * 1e-6)
                del po, pr, pt, pu
            # the agent's policy step on this observation width (actor -> sample -> log-prob + critic value) as ONE launch
            # (wl_actor_critic_act: the first layers as skinny fp32 GEMMs on the matrix pipe)
            from wheeledlab_amd.policy import ActorCritic as KernelAC
            kac = KernelAC(t.OBS_DIM, 2, "relu" if name == "elevation" else "elu", device=dev, seed=0)
            ob = torch.randn(n, t.OBS_DIM, device=dev)
            pa, pm = torch.empty(n, 2, device=dev), torch.empty(n, 2, device=dev)
            pl, pv = torch.empty(n, device=dev), torch.empty(n, device=dev)
            for i in range(10):
                kac.act(ob, pa, pm, pl, pv, 42, i)
            torch.cuda.synchronize()
            q0.record()
            for i in range(100):
                kac.act(ob, pa, pm, pl, pv, 42, i)
            q1.record()
            torch.cuda.synchronize()
            pus = q0.elapsed_time(q1) * 1e3 / 100
            flop = 2.0 * n * (2 * t.OBS_DIM * 64 + 2 * 64 * 64 + 3 * 64)
            other[name]["policy_step_us"] = pus
            other[name]["policy_step_TFLOPs"] = flop / (pus * 1e-6) / 1e12
            if name == "elevation" and n <= 32768:
                # the runner's collection loop (actor -> sample -> env.step -> storage rows) as ONE persistent launch
                # (wl_elev_collect_rollout: actor layer 1 in the blocks' registers, observation rows in LDS)
                from wheeledlab_amd.policy import RolloutStorage
                kc = 32
                cst = RolloutStorage(kc, n, t.OBS_DIM, 2, dev)
                cst.observations[0].copy_(t.observe())
                kac.planes = False
                t.collect_rollout(kac, cst)
                torch.cuda.synchronize()
                q0.record()
                for _ in range(4):
                    t.collect_rollout(kac, cst)
                q1.record()
                torch.cuda.synchronize()
                cus = q0.elapsed_time(q1) * 1e3 / (4 * kc)
                other[name]["collect_rollout_us_per_step"] = cus
                other[name]["collect_rollout_env_steps_per_s"] = n / (cus * 1e-6)
                del cst
            del t, a, kac, ob

    # secondary: BASELINE.json configs[4]'s named kernel -- the depth ray-cast of the visual task's camera against the
    # heightfield: n cars of the elevation task standing on the synthetic 800 x 800 terrain after 8 steps, one 60 x 80
    # distance_to_image_plane image each (clipping range 100 m, visual/mushr_visual_env_cfg.py:240)
    if rank == 0 and world == 1 and not args.headline_only:
        from wheeledlab_amd.core import DepthCamera, ElevBatch
        t = ElevBatch(n, device=dev, seed=42)
        t.reset()
        t.rollout(torch.rand(8, n, 2, device=dev) * 2 - 1)
        cam = DepthCamera(t.hf, dev)
        img = torch.empty(n, 60, 80, device=dev)
        for _ in range(3):
            cam.render(t, 100.0, img)
        torch.cuda.synchronize()
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dus = 1e30
        for _ in range(3):
            q0.record()
            for _ in range(8):
                cam.render(t, 100.0, img)
            q1.record()
            torch.cuda.synchronize()
            dus = min(dus, q0.elapsed_time(q1) * 1e3 / 8)
        hit = float((img < 100.0).float().mean())
        other["visual_depth"] = {"us_per_render": dus, "images_per_s": n / (dus * 1e-6), "rays_per_s": n * 4800 / (dus * 1e-6),
                                 "image": "60 x 80 fp32 distance_to_image_plane", "hit_fraction": hit,
                                 "workload": f"{n} elevation-task cars on the synthetic 800 x 800 heightfield (0.05 m), max depth 100 m",
                                 "roofline": roofline_block("depth", n, dus, "visual_depth_tile_kernel",
                                                            "valu issue + divergence (max-pyramid walk; the image write is the only HBM stream)",
                                                            profile_kernels=[("visual_depth_tile_kernel<0>",)] if n == ENVS_PER_GPU else None)}
        # this kernel's governing roofline is the VALU pipe, not HBM: instructions issued x 2 cycles / (1024 SIMDs x shader cycles)
        dc = (other["visual_depth"]["roofline"].get("counters") or {}).get("visual_depth_tile_kernel") or {}
        other["visual_depth"]["roofline"]["valu_frac"] = dc.get("valu_pipe_frac")
        other["visual_depth"]["roofline"]["governing"] = "valu_frac (HBM fraction reported for the contract; the walk is instruction-bound)"
        if not args.no_cpu_baseline:
            try:
                # the CPU beside it: the oracle's exact cell-by-cell intersection (oracle/depth.c, double precision, OpenMP over the
                # images) on this box's host cores, on the first 512 of the same poses
                from oracle import depth as OD
                from oracle import visual_step as OV
                m = min(n, 512)
                st = t.state[:, :m].cpu().numpy()
                field = (t.height.cpu().numpy(), float(t._hf.x0), float(t._hf.y0), float(t._hf.cell))
                OD.depth(OV.visual_params(), st[0:3, :8].T, st[3:7, :8].T, field, 100.0)        # builds / loads the library
                c0 = time.perf_counter()
                OD.depth(OV.visual_params(), st[0:3].T, st[3:7].T, field, 100.0)
                c1 = time.perf_counter() - c0
                other["visual_depth"]["cpu_baseline"] = {"value": m * 4800 / c1, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
                                                         "sample": f"{m} images of the same poses through oracle/depth.c (exact per-cell "
                                                                   f"intersection, double precision, OpenMP), {c1:.2f} s"}
            except Exception as ex:   # noqa: BLE001 -- building / loading oracle/depth.c needs make + gcc + OpenMP on the box: a
                other["visual_depth"]["cpu_baseline"] = {"error": repr(ex)}   # secondary figure must never cost the headline line
        del t, cam, img

    # secondary: BASELINE.json configs[4] AS A TASK -- the visual task stepped on the heightfield terrain with the depth image as the
    # policy observation (extension id Isaac-MushrVisualDepthRL-v0): two launches per env.step() (step on the heightfield, depth
    # ray-cast into the observation rows), far plane 20 m as the task's camera is configured
    if rank == 0 and world == 1 and not args.headline_only:
        from wheeledlab_amd.core import VisualDepthBatch
        t = VisualDepthBatch(n, device=dev, seed=42)
        t.reset()
        k = 16
        a = torch.rand(k, n, 2, device=dev) * 2 - 1
        a[:, :, 0] = a[:, :, 0].abs()
        for _ in range(3):
            t.rollout(a)
        torch.cuda.synchronize()
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        us = 1e30
        for _ in range(3):
            q0.record()
            for _ in range(4):
                t.rollout(a)
            q1.record()
            torch.cuda.synchronize()
            us = min(us, q0.elapsed_time(q1) * 1e3 / (4 * k))
        t.observe()
        torch.cuda.synchronize()
        q0.record()
        for _ in range(16):
            t.observe()
        q1.record()
        torch.cuda.synchronize()
        obs_us = q0.elapsed_time(q1) * 1e3 / 16
        other["visual_depth_task"] = {
            "us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "obs_dim": t.OBS_DIM, "render_us": obs_us, "step_launch_us": us - obs_us,
            "launches": "two per env.step() (visual_step_kernel<HeightFieldGround>, visual_depth_tile_kernel)",
            "workload": f"Isaac-MushrVisualDepthRL-v0 (extension): {n} envs on the synthetic 800 x 800 heightfield, 80 x 80 traversability map, "
                        f"depth image 60 x 80 clipped at {t.max_depth:g} m as observation",
            "hit_fraction": float((t.obs[:, :4800] < t.max_depth).float().mean()),
            "roofline": roofline_block("visual_depth", n, us, "visual_step_kernel<HeightFieldGround> + visual_depth_tile_kernel",
                                       "valu issue + divergence (the depth walk) + latency (40 dependent sub-steps with terrain gathers)",
                                       profile_kernels=[("visual_step_kernel<", "HeightFieldGround>("), ("visual_depth_tile_kernel<1>",)]
                                       if n == ENVS_PER_GPU else None),
            "test": "tests/test_gpu_visual_depth_task.py"}
        del t, a

    # secondary: the same workload driven step by step through the drop-in Python surface
    # (registry.make -> ClipAction -> RslRlVecEnvWrapper.step), i.e. what a Python RL loop sees per env.step() call
    py_rate = None
    if rank == 0 and world == 1 and not args.headline_only:
        from wheeledlab_amd import registry, tasks  # noqa: F401
        from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
        cfg = registry.parse_env_cfg("Isaac-MushrDriftRL-v0", device=str(dev), num_envs=n)
        e = registry.make("Isaac-MushrDriftRL-v0", cfg=cfg)
        e.action_space.low, e.action_space.high = -1.0, 1.0
        w = RslRlVecEnvWrapper(ClipAction(e))
        # the host outruns the GPU here, and while the launch queue deepens for the first time the HIP runtime grows its
        # signal / kernarg pools (measured: ~80 us per call for the first ~1000 calls of a process, 10.6 us after):
        # warm up past that, then time the steady state
        for i in range(1536):
            w.step(actions[i % ROLLOUT])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(2048):
            w.step(actions[i % ROLLOUT])
        torch.cuda.synchronize()
        py_rate = n * 2048 / (time.perf_counter() - t1)
        del w, e

    # secondary: the SAME fused step at env counts where it is throughput- rather than launch-bound (SURVEY 8(d) config 2:
    # "also sweep N ... to expose the bandwidth-bound regime").  Run in a fresh process: with the GB-sized buffers carved
    # out of this process's caching-allocator leftovers the same launches measured 12-15 % slower (round 2, docs/HISTORY.md)
    sweep, other_sweep = [], []
    if rank == 0 and world == 1 and not args.no_sweep:
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--sweep-child"], capture_output=True, text=True,
                                 timeout=600, check=True).stdout.strip().splitlines()[-1]
            both = json.loads(out)
            sweep, other_sweep = both["drift"], both["other"]
        except Exception as ex:   # noqa: BLE001 -- the sweep is a secondary figure: never lose the headline line over it
            print(f"[bench] sweep subprocess failed ({ex!r}); measuring in-process", file=sys.stderr, flush=True)
            sweep = large_n_sweep(dev)

    if rank == 0:
        total_envs = n * world
        # the same fraction from the committed artefacts: algorithmic bytes / the kernel's average duration in the rocprofv3
        # summary of this script (profiles/), and counter bytes / duration in the PMC passes
        prof = rocprof_avg_us("drift_step_kernel")
        frac_profile = BYTES_PER_ENV_STEP * n / (prof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS if prof and n == ENVS_PER_GPU else None
        pe = pmc_entry("drift", n)
        frac_counters = pe.get("frac_counters") if pe and not pe["stale"] else None
        detail = {
            "metric": "env steps/sec (whole node), drift task @ 4096 envs/GPU",
            "value": total_envs * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RSS_DRIFT_CONFIG drift task, {n} envs/GPU, flat terrain, fused dynamics+mdp HIP "
                                   f"kernel, U(-1,1) actions pre-staged in HBM, {ROLLOUT}-step rollout storage",
                       "envs_per_gpu": n, "total_envs": total_envs, "decimation": 4, "sim_dt": 0.005,
                       "parallelism": f"env-shard x{world}, metric all-reduce / {every} steps"},
            "gpu_event_ms_per_step": gpu_ms / args.steps,
            "timing": {"repeats": REPEATS, "statistic": "median of the per-block max-over-ranks wall time",
                       "block_ms": [round(w * 1e3, 4) for w in walls],
                       # every block counted, incl. the ones the metric all-reduce falls into (the median skips them at K < 128)
                       "mean_value": total_envs * args.steps / wall_mean, "mean_ms_per_step": wall_mean * 1e3 / args.steps,
                       "metric_reductions_in_timed_blocks": reductions_in_timed_blocks, "allreduce_every": every},
            "rccl": rccl,
            # `bound` names the roofline `peak` belongs to (the contract: "hbm" | "mfma"); `regime` says what actually limits the
            # launch at this size
            "roofline": {"bound": "hbm", "regime": regime, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
                         "frac_is": "bytes_per_env_step x envs_per_launch / ms_per_step / peak",
                         "frac_launch": achieved_launch / HBM_PEAK_GBS,
                         # SURVEY.md 8(d)'s own per-unit figure (270 B: it leaves out the episode-sum rows and the flag bytes the
                         # kernel also moves) beside the 334 B this file counts row by row
                         "bytes_per_env_step_survey": BYTES_PER_ENV_STEP_SURVEY,
                         "frac_survey": BYTES_PER_ENV_STEP_SURVEY * n / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "profile_us": prof["avg_us"] if prof else None,
                         "valu_frac": sq.get("valu_pipe_frac") if sq else None, "sq_counters": sq,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "kernel": "drift_step_kernel<FlatGround>", "launch_us": launch_us,
                         "frac_uses": "ms_per_step (wall clock of the timed blocks); frac_launch: launch_us (HIP events on the launch stream)",
                         "rocprof_kernel_stats": prof, "frac_profile": frac_profile, "frac_counters": frac_counters,
                         "wasted_traffic": traffic / (BYTES_PER_ENV_STEP * n) if traffic else None,
                         "bytes_per_env_step": BYTES_PER_ENV_STEP, "envs_per_launch": n},
            "episode_metrics": {"resets": float(metric_sum[8]), "timeouts": float(metric_sum[9]),
                                "out_of_bounds": float(metric_sum[10]), "nonfinite": float(metric_sum[14])},
        }
        detail["python_surface_env_steps_per_s"] = py_rate
        detail["persistent_rollout"] = persistent
        detail["policy_rollout"] = policy
        detail["training_iteration"] = train
        detail["other_tasks"] = other
        if sweep:
            detail["large_n_sweep"] = sweep
        if other_sweep:
            detail["other_tasks_large_n"] = other_sweep
        if world == 1 and not args.no_cpu_baseline:
            detail["cpu_baseline"] = cpu_baseline(n)
        # The driver parses the LAST stdout line; round 4's grew to 21.9 KB and was not parsed.  Everything measured goes to a
        # side file (and, on request, to an earlier stdout line that does not start with `{`); the one JSON line is the compact
        # digest of it, asserted below to stay under 6 KB.
        detail_path = write_detail(detail, args.detail_out)
        line = compact_line(detail, detail_path)
        text = json.dumps(line, separators=(",", ":"))
        # never lose the headline over a secondary section: should the line ever outgrow the limit, the optional digests go first
        # (they stay in the side file), and the test suite fails on the size (tests/test_gpu_bench_contract.py)
        for optional in ("other_tasks_large_n", "large_n_sweep", "other_tasks", "episode_metrics"):
            if len(text) < MAX_LINE_BYTES:
                break
            print(f"[bench] line is {len(text)} bytes (limit {MAX_LINE_BYTES}): dropping `{optional}` from it", file=sys.stderr, flush=True)
            line.pop(optional, None)
            text = json.dumps(line, separators=(",", ":"))
        if args.print_detail:
            print("[bench detail] " + json.dumps(detail), flush=True)
        print(text, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
