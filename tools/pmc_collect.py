#!/usr/bin/env python3
"""Fold rocprofv3 PMC passes of the forward (and backward) launch into profiles/pmc_traffic.json.

On the GPU box (each counter set in its own run, kernel trace only -- never with sys/hip traces):
    tools/pmc_run.sh fetch  "FETCH_SIZE"               fwd --steps 20 [--kappa K]
    tools/pmc_run.sh write  "WRITE_SIZE"               fwd --steps 20 [--kappa K]
    tools/pmc_run.sh hit    "TCC_HIT_sum TCC_MISS_sum" fwd --steps 20 [--kappa K]
    tools/pmc_run.sh sq     "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" fwd --steps 20
    (the same four with `bwd` instead of `fwd` for the backward launch: tags bfetch, bwrite, bhit, bsq)
then here:  python tools/pmc_collect.py gpurun_out/pmc_fetch gpurun_out/pmc_write ... [--key c3:kappa=0.5:binned]
Units: FETCH_SIZE / WRITE_SIZE are KiB (MI355X_MICROARCH.md "HBM"); on gfx950 FETCH_SIZE counts wide
reads at half their size, so the corrected figure is (2 x FETCH_SIZE + WRITE_SIZE) KiB.
The record carries the hash of the kernel sources it was measured on (pigs_amd.build.source_hash):
bench.py reports the traffic only while that hash still matches.
"""
import collections, csv, glob, importlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
key = "c3:kappa=0.5:binned"
if "--key" in sys.argv:
    key = sys.argv[sys.argv.index("--key") + 1]
    args.remove(key)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    # gpurun merges every collection's raw files into the same local directory: only the newest pass counts
    found = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in found[-1:]:
        per_dispatch = collections.defaultdict(float)     # (dispatch, kernel, counter) -> sum over instances
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per_dispatch.items():
            if "pigs::" in k:
                acc[k.split("(")[0].replace("void ", "")][c].append(v)
allk = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
fw = next((v for k, v in allk.items() if "tile_forward_kernel<1, 7>" in k), None)
if fw is None or "FETCH_SIZE" not in fw or "WRITE_SIZE" not in fw:
    raise SystemExit("no forward-kernel FETCH_SIZE / WRITE_SIZE found in " + " ".join(args))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
N, M = 65536, 1 << 20
table[key] = {
    "kernel": "tile_forward_kernel<1,7>",
    "source_hash": importlib.import_module("pigs_amd.build").source_hash(),
    "FETCH_SIZE_KiB": fw["FETCH_SIZE"], "WRITE_SIZE_KiB": fw["WRITE_SIZE"],
    "hbm_bytes_raw": (fw["FETCH_SIZE"] + fw["WRITE_SIZE"]) * 1024,
    "hbm_bytes_per_launch": (2 * fw["FETCH_SIZE"] + fw["WRITE_SIZE"]) * 1024,
    "algorithmic_bytes": 24 * N + 36 * M,
    "TCC_HIT_sum": fw.get("TCC_HIT_sum"), "TCC_MISS_sum": fw.get("TCC_MISS_sum"),
    "SQ": {c: v for c, v in fw.items() if c.startswith("SQ_")},
    "how": "tools/pmc_collect.py over separate rocprofv3 --kernel-trace --pmc passes of `tools/prof_step.py fwd --steps 20` "
           "(the forward launch alone on one plan; FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum+TCC_MISS_sum; the SQ set); mean over "
           "the launches of the kernel. Units KiB (MI355X_MICROARCH.md 'HBM'). gfx950 FETCH_SIZE halves wide reads: "
           "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB is the corrected upper figure, hbm_bytes_raw the "
           "uncorrected one.",
}
bw = next((v for k, v in allk.items() if "tile_backward_kernel<1, 7>" in k), None)
un = next((v for k, v in allk.items() if "plan_unpermute_kernel<1>" in k), None)
if bw is not None and "FETCH_SIZE" in bw and "WRITE_SIZE" in bw:
    un = un or {}
    table[key]["backward"] = {
        "kernel": "tile_backward_kernel<1,7> + plan_unpermute_kernel<1>",
        "FETCH_SIZE_KiB": bw["FETCH_SIZE"] + un.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB": bw["WRITE_SIZE"] + un.get("WRITE_SIZE", 0.0),
        "hbm_bytes_per_launch": (2 * (bw["FETCH_SIZE"] + un.get("FETCH_SIZE", 0.0)) + bw["WRITE_SIZE"] + un.get("WRITE_SIZE", 0.0)) * 1024,
        "algorithmic_bytes": 48 * N + 36 * M,
        "TCC_HIT_sum": bw.get("TCC_HIT_sum"), "TCC_MISS_sum": bw.get("TCC_MISS_sum"),
        "SQ": {c: v for c, v in bw.items() if c.startswith("SQ_")},
        "how": "the same passes over `tools/prof_step.py bwd --steps 20` (the backward launches alone on one plan); both kernels "
               "of the backward summed; same units and gfx950 correction as the forward record",
    }
# only the kernels of THIS collection (records of earlier rounds' kernels do not linger)
table["all_kernels"] = allk
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(table[key], indent=1))
