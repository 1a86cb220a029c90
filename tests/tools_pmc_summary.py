"""Summarise the CSVs written by tests/tools_pmc.sh into profiles/<tag>_pmc_summary.json (per-launch means)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
for name in ["sq1", "sq2", "fetch", "write", "calfetch", "calwrite"]:
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{name}", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        kn = r["Kernel_Name"]
        if "sqp_kernel" in kn or "stream_rw" in kn:
            agg[(kn.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, cn), v in agg.items():
        out.setdefault(name, {})[f"{cn} [{kn}]"] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
