"""Digest of tools/r02_profile.sh: profiles/r02_pmc_traffic.json (HBM bytes per launch of the fused drift step, with the
FETCH_SIZE / WRITE_SIZE factors calibrated on known bytes in the same access width), profiles/r02_pmc_sq.json (SQ
counters: VALU instructions per wavefront, wait / issue-stall / VALU-active shares, VALU-pipe occupancy estimate) and
the kernel-statistics CSVs.  usage: r02_pmc_report.py <gpurun_out/r02>"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

O = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = O   # (gpurun merges only gpurun_out/ back: the files are copied into profiles/ by hand)


def counters(d, kernel_substr):
    acc, cnt, dur = defaultdict(float), defaultdict(int), []
    for path in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel_substr not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: acc[k] / cnt[k] for k in acc}, (sum(dur) / len(dur) if dur else None), (max(cnt.values()) if cnt else 0)


# ---- calibration: layout_bw launches stream_kernel<34,30,false> (SoA dword) on n = 1 M and 4 M, copy4_kernel (16 B / lane)
calib = {}
for name, sub, rd, wr in (("soa_dword_4M", "stream_kernel<34, 30, false>", 34 * 4 * 4194304, 30 * 4 * 4194304),
                          ("float4_copy", "copy4_kernel", 20 * 4 * 4194304, 20 * 4 * 4194304)):
    f, _, nf = counters("calib_FETCH", sub)
    w, _, nw = counters("calib_WRITE", sub)
    # the SoA kernel runs at two sizes: keep the dispatches of the larger one (counter value above half the maximum)
    def big(d, key):
        vals = []
        for path in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if sub in r["Kernel_Name"] and r["Counter_Name"] == key:
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            return None
        m = max(vals)
        sel = [v for v in vals if v > 0.6 * m]
        return sum(sel) / len(sel)
    fk, wk = big("calib_FETCH", "FETCH_SIZE"), big("calib_WRITE", "WRITE_SIZE")
    if fk and wk:
        calib[name] = {"known_read_bytes": rd, "known_write_bytes": wr, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                       "fetch_factor": rd / (fk * 1024), "write_factor": wr / (wk * 1024)}
ff = calib.get("soa_dword_4M", {}).get("fetch_factor", 2.0)
wf = calib.get("soa_dword_4M", {}).get("write_factor", 1.0)

BYTES_R, BYTES_W = (23 + 4 + 7) * 4 + 4 + 8, (23 + 7) * 4 + 4 + 14 * 4 + 4 + 2
traffic = {"_doc": "HBM traffic of drift_step_kernel per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes; workload "
                   "tools/pmc_run.py = bench.py's launch: outputs into rollout storage, no int64 dones row).  Counters are KiB.  "
                   "MI355X_MICROARCH.md calibrates FETCH_SIZE x2 for 16-byte-per-lane loads and calls other widths uncalibrated: the "
                   "factors here are measured on tools/microbench/layout_bw (known bytes, this kernel's 4-byte buffer loads / stores, "
                   "34 rows in / 30 rows out at 4 M envs; a float4 copy beside it) in the same profiling pass.",
           "calibration": calib, "fetch_factor_used": ff, "write_factor_used": wf, "entries": {}}
for n in (4096, 4194304):
    f, _, _ = counters(f"traffic_FETCH_{n}", "drift_step")
    w, _, _ = counters(f"traffic_WRITE_{n}", "drift_step")
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        rb, wb = f["FETCH_SIZE"] * 1024 * ff, w["WRITE_SIZE"] * 1024 * wf
        alg = (BYTES_R + BYTES_W) * n
        traffic["entries"][str(n)] = {"fetch_size_kib_raw": f["FETCH_SIZE"], "write_size_kib_raw": w["WRITE_SIZE"], "read_bytes": rb,
                                      "write_bytes": wb, "traffic_bytes": rb + wb, "algorithmic_bytes": alg, "ratio": (rb + wb) / alg,
                                      "read_ratio": rb / (BYTES_R * n), "write_ratio": wb / (BYTES_W * n)}
json.dump(traffic, open(os.path.join(P, "r02_pmc_traffic.json"), "w"), indent=1)

sq = {"_doc": "SQ counters of drift_step_kernel per launch (rocprofv3 --pmc, two passes; tools/pmc_run.py workload).  SQ_* time counters "
              "are quad-cycles per wavefront summed over wavefronts.  valu_pipe_frac: share of the kernel's duration the VALU pipes "
              "would be busy if every VALU instruction took 2 cycles of its SIMD (4 for the ~9 % transcendental / 64-bit-multiply "
              "ones are not separated by the counters: a lower bound), = SQ_INSTS_VALU x 2 / (1024 SIMDs x shader cycles of the "
              "launch, from GRBM_GUI_ACTIVE / 8 XCDs).", "entries": {}}
for n in (4096, 4194304):
    a, dur_a, _ = counters(f"sq_a_{n}", "drift_step")
    b, dur_b, _ = counters(f"sq_b_{n}", "drift_step")
    if not a:
        continue
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    e = {"kernel_ns_under_pmc": dur_a, "waves": a.get("SQ_WAVES"), "valu_insts_per_wave": a["SQ_INSTS_VALU"] / a["SQ_WAVES"],
         "wave_cycles_per_wave_quadcycles": wc / a["SQ_WAVES"],
         "share_of_wave_cycles": {k: a[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in a},
         "salu_per_wave": b.get("SQ_INSTS_SALU", 0) / a["SQ_WAVES"], "vmem_rd_per_wave": b.get("SQ_INSTS_VMEM_RD", 0) / a["SQ_WAVES"],
         "vmem_wr_per_wave": b.get("SQ_INSTS_VMEM_WR", 0) / a["SQ_WAVES"], "lds_per_wave": b.get("SQ_INSTS_LDS", 0) / a["SQ_WAVES"]}
    if b.get("GRBM_GUI_ACTIVE"):
        cyc = b["GRBM_GUI_ACTIVE"] / 8.0
        e["shader_cycles_per_launch"] = cyc
        e["valu_pipe_frac"] = a["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
    sq["entries"][str(n)] = e
json.dump(sq, open(os.path.join(P, "r02_pmc_sq.json"), "w"), indent=1)

for src, dst in (("bench_stats", "r02_bench_kernel_stats.csv"), ("tasks_stats", "r02_tasks_kernel_stats.csv")):
    fs = glob.glob(f"{O}/{src}/**/*kernel_stats.csv", recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(O, dst))
for tag in ("traffic_FETCH_4096", "traffic_WRITE_4096", "traffic_FETCH_4194304", "traffic_WRITE_4194304", "sq_a_4096", "sq_b_4096",
            "sq_a_4194304", "sq_b_4194304", "calib_FETCH", "calib_WRITE"):
    fs = glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(O, f"r02_pmc_{tag}.csv"))
print(json.dumps({"traffic": traffic["entries"], "calibration": calib, "sq": sq["entries"]}, indent=1))
