"""Summarise rocprofv3 --pmc counter CSVs: mean counter value per dispatch of the fused step kernel."""
import csv, glob, json, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        vals = {}
        for r in csv.DictReader(open(f)):
            if "drift_step_kernel" not in r.get("Kernel_Name", ""):
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            out[f"{d.rstrip('/').split('/')[-1]}:{k}"] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
