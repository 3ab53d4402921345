"""Summarise rocprofv3 --pmc counter_collection.csv files per stage kernel (mean per dispatch)."""
import csv, glob, sys, collections, json
import numpy as np
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n_ = r["Kernel_Name"]
            k = ("fwd_persist" if "k_fwd_persist" in n_ else "adj_persist" if "k_adj_persist" in n_ else
                 "fwd" if ("k_fwd_stage" in n_ or "k_fwd_tile" in n_) else "adj" if ("k_adj_stage" in n_ or "k_adj_tile" in n_) else None)
            if k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {})[c] = float(np.mean(v))
print(json.dumps(out, indent=1))
