"""Per-kernel means of the counters of rocprofv3 --pmc output directories (summed over the counter's
instances per dispatch, averaged over the dispatches of a kernel)."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            if "pigs::" in k:
                acc[k.split("(")[0].replace("void ", "")][c].append(v)
    for k, cs in sorted(acc.items()):
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in sorted(cs.items())})
print(json.dumps(out, indent=1))
