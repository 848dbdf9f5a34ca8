"""Digest of tools/profile_pmc.sh -> <out>/<tag>_pmc.json: per (task, env count) and kernel the HBM bytes per launch (FETCH_SIZE /
WRITE_SIZE passes, factors calibrated on the known bytes of tools/microbench/pmc_calib.hip in the same pass -- `factors_from` says
whether that happened -- else the guide's: FETCH x2, WRITE x1), the kernel's duration under the counters, and the SQ view (VALU instructions per wavefront, wait /
issue-stall / active shares, VALU-pipe occupancy).  Stamped with the fingerprint of csrc/ + the header so that bench.py can
tell whether the counters belong to the library it loaded.    usage: pmc_report.py <gpurun_out/tag> <tag>"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

O = sys.argv[1]
TAG = sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import ALGO_BYTES, csrc_fingerprint  # noqa: E402

KERNELS = {"drift": ["drift_step_kernel"], "elev": ["elev_step_scan_kernel", "elev_step_kernel", "elev_scan_kernel", "elev_scan_lds_kernel"],
           "visual": ["visual_step_kernel", "visual_obs_kernel"], "depth": ["visual_depth_kernel", "visual_depth_tile_kernel"],
           "visual_depth": ["visual_step_kernel", "visual_depth_tile_kernel"]}


def rows(d):
    for path in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        yield from csv.DictReader(open(path))


def base_name(k):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(\w+)", k)
    return m.group(1) if m else k


def counters(d, kernel):
    """mean counter values per dispatch of `kernel` (base name) + mean duration + dispatch count.  The first dispatch of a
    workload is its warm-up (cold caches, first-touch page faults): dropped when there are at least three."""
    per = defaultdict(lambda: defaultdict(float))   # dispatch id -> counter -> value
    dur = {}
    for r in rows(d):
        if base_name(r["Kernel_Name"]) != kernel:
            continue
        i = int(r["Dispatch_Id"])
        per[i][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[i] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(per)
    if len(ids) >= 3:
        ids = ids[1:]
    if not ids:
        return {}, None, 0
    names = set().union(*[per[i].keys() for i in ids])
    return {c: sum(per[i][c] for i in ids) / len(ids) for c in names}, sum(dur[i] for i in ids) / len(ids), len(ids)


def calibration():
    """FETCH_SIZE / WRITE_SIZE (KiB) against the known bytes of tools/microbench/pmc_calib.hip's two kernels, each counter in its own
    pass: fetch_factor = known read bytes / (FETCH_SIZE x 1024), write_factor likewise.  The guide's figures are x2 and x1 (gfx950
    counts 64-byte fetch units in a counter documented in 32-byte units); round 2 measured 1.9999 / 1.000."""
    cal = {}
    if not os.path.isdir(f"{O}/calib_FETCH"):
        return cal
    for name, kernel, rd, wr in (("soa_dword_4M", "calib_soa_dword_kernel", 34 * 4 * 4194304, 30 * 4 * 4194304),
                                 ("float4_copy_1GiB", "calib_copy4_kernel", 1 << 30, 1 << 30)):
        f, _, nf = counters("calib_FETCH", kernel)
        w, _, nw = counters("calib_WRITE", kernel)
        if f.get("FETCH_SIZE") and w.get("WRITE_SIZE"):
            cal[name] = {"known_read_bytes": rd, "known_write_bytes": wr, "FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"],
                         "dispatches": min(nf, nw), "fetch_factor": rd / (f["FETCH_SIZE"] * 1024), "write_factor": wr / (w["WRITE_SIZE"] * 1024)}
    return cal


cal = calibration()
# the factors applied below: the measured ones of the drift step's own access shape when this pass calibrated, else the guide's
ff = cal.get("soa_dword_4M", {}).get("fetch_factor", 2.0)
wf = cal.get("soa_dword_4M", {}).get("write_factor", 1.0)
factors_from = "calibration.soa_dword_4M (measured in this pass)" if "soa_dword_4M" in cal else "MI355X_MICROARCH.md (x2 / x1): no calibration in this pass"
out = {"_doc": "rocprofv3 --pmc digests per (task:envs) and kernel, per launch.  traffic = FETCH_SIZE KiB x 1024 x fetch_factor + "
               "WRITE_SIZE KiB x 1024 x write_factor (separate passes).  SQ_* time counters are quad-cycles per wavefront summed "
               "over wavefronts; valu_pipe_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x shader cycles of the launch, "
               "GRBM_GUI_ACTIVE / 8 XCDs): a lower bound (transcendentals take 4+).  duration_ns is the kernel's mean duration "
               "with counters attached (slower than untraced).  Workload: tools/pmc_run.py.",
       "csrc_fingerprints": {t: csrc_fingerprint(t) for t in KERNELS}, "calibration": cal, "fetch_factor_used": ff, "write_factor_used": wf, "factors_from": factors_from, "entries": {}}
tags = sorted({os.path.basename(p)[len("FETCH_"):] for p in glob.glob(f"{O}/FETCH_*") if os.path.isdir(p)})
for tag in tags:
    task, n = tag.rsplit("_", 1)
    n = int(n)
    entry = {"n_envs": n, "algorithmic_bytes": ALGO_BYTES.get(task, 0) * n, "kernels": {}}
    tot, dur_tot = 0.0, 0.0
    for k in KERNELS.get(task, []):
        f, dur_f, nf = counters(f"FETCH_{tag}", k)
        w, dur_w, _ = counters(f"WRITE_{tag}", k)
        a, dur_a, na = counters(f"sqa_{tag}", k)
        b, dur_b, _ = counters(f"sqb_{tag}", k)
        if not (f or a):
            continue
        e = {"dispatches": nf or na}
        if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
            e["read_bytes"], e["write_bytes"] = f["FETCH_SIZE"] * 1024 * ff, w["WRITE_SIZE"] * 1024 * wf
            e["traffic_bytes"] = e["read_bytes"] + e["write_bytes"]
            tot += e["traffic_bytes"]
            # the counters' own clock: bytes the counters saw / the kernel's mean duration in the two traffic passes
            e["duration_traffic_passes_ns"] = 0.5 * (dur_f + dur_w)
            e["frac_counters"] = e["traffic_bytes"] / (e["duration_traffic_passes_ns"] * 1e-9) / 8e12
            dur_tot += e["duration_traffic_passes_ns"]
        if a and a.get("SQ_WAVES"):
            wc = a.get("SQ_WAVE_CYCLES", 0.0)
            e.update({"duration_ns": dur_a, "waves": a["SQ_WAVES"], "valu_insts_per_wave": a["SQ_INSTS_VALU"] / a["SQ_WAVES"],
                      "wave_cycles_per_wave_quadcycles": wc / a["SQ_WAVES"],
                      "share_of_wave_cycles": {c: a[c] / wc for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                                                                     "SQ_WAIT_INST_ANY") if c in a and wc}})
            if b:
                e.update({"salu_per_wave": b.get("SQ_INSTS_SALU", 0) / a["SQ_WAVES"], "vmem_rd_per_wave": b.get("SQ_INSTS_VMEM_RD", 0) / a["SQ_WAVES"],
                          "vmem_wr_per_wave": b.get("SQ_INSTS_VMEM_WR", 0) / a["SQ_WAVES"], "lds_per_wave": b.get("SQ_INSTS_LDS", 0) / a["SQ_WAVES"]})
                if b.get("GRBM_GUI_ACTIVE"):
                    cyc = b["GRBM_GUI_ACTIVE"] / 8.0
                    e["shader_cycles_per_launch"] = cyc
                    e["valu_pipe_frac"] = a["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
        entry["kernels"][k] = e
    if tot:
        entry["traffic_bytes"] = tot
        entry["frac_counters"] = tot / (dur_tot * 1e-9) / 8e12     # all kernels of the step: counter bytes / summed durations / 8 TB/s
        if entry["algorithmic_bytes"]:
            entry["ratio"] = tot / entry["algorithmic_bytes"]
    out["entries"][f"{task}:{n}"] = entry
json.dump(out, open(os.path.join(O, f"{TAG}_pmc.json"), "w"), indent=1)
fs = glob.glob(f"{O}/bench_stats/**/*kernel_stats.csv", recursive=True)
if fs:
    # rocprofv3 writes one summary per process (bench.py's sweep runs in a child): the bench process itself is the one with the
    # headline kernel's thousands of launches; the sweep child's summary is kept beside it
    def headline_calls(path):
        return sum(int(r["Calls"]) for r in csv.DictReader(open(path)) if "drift_step_kernel" in r.get("Name", ""))
    fs.sort(key=headline_calls, reverse=True)
    shutil.copy(fs[0], os.path.join(O, f"{TAG}_bench_kernel_stats.csv"))
    if len(fs) > 1:
        shutil.copy(fs[1], os.path.join(O, f"{TAG}_bench_sweep_kernel_stats.csv"))
print(json.dumps(out, indent=1))
