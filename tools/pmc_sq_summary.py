"""Summarise rocprofv3 --pmc SQ passes of the fused drift step (tools/pmc_run.py workload): per-launch averages of the
raw counters and the fractions of wavefront time they imply (SQ_* time counters are in quad-cycles; WAIT_ANY +
WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, /opt/skills/guides/MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import csv, glob, json, sys
from collections import defaultdict

out = {}
for tag in sys.argv[2:]:
    acc, cnt, dur = defaultdict(float), defaultdict(int), []
    for path in glob.glob(f"{sys.argv[1]}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "drift_step" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
            if r["Counter_Name"] == "SQ_WAVES":
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    avg = {k: acc[k] / cnt[k] for k in acc}
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    s = {"launches": max(cnt.values()) if cnt else 0, "per_launch": avg,
         "kernel_ns_under_pmc": sum(dur) / len(dur) if dur else None}
    if wc:
        s["fraction_of_wave_cycles"] = {k: avg[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                                                                  "SQ_WAIT_INST_ANY") if k in avg}
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg and avg["SQ_WAVES"]:
        s["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
    if "GRBM_GUI_ACTIVE" in avg and "SQ_ACTIVE_INST_VALU" in avg and avg["GRBM_GUI_ACTIVE"]:
        # gfx94x derived-metric formula (no gfx950 section ships with ROCm 7.2): 4 quad-cycles -> cycles, 1024 SIMDs
        s["VALUBusy_pct_gfx94x_formula"] = 100.0 * avg["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / avg["GRBM_GUI_ACTIVE"]
    out[tag] = s
print(json.dumps(out, indent=1))
