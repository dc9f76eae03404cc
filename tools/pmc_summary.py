"""Averages rocprofv3 --pmc counter_collection CSVs per launch of the dominant kernel."""
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for row in csv.DictReader(open(path)):
            if not row["Kernel_Name"].startswith(("k_fused<", "void k_fused<", "k_wf_extend", "void k_wf_extend")):
                continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, by_dispatch in per.items():
            vals = list(by_dispatch.values())
            out[name] = sum(vals) / len(vals)
            out["_launches"] = len(vals)
print(json.dumps(out, indent=1, sort_keys=True))
