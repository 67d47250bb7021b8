"""Per-kernel HBM bytes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; values in KiB).

    python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json

FETCH_SIZE on gfx950 reports half of the bytes of a streaming read (MI355X_MICROARCH.md, HBM
section; calibrated in profiles/r01_pmc_score_1024x2048.json), hence the factor 2.
"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if name.startswith("mn_"):
                acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python "
                 "tools/prof_components.py 6 (Infinity Cache evicted before every image)",
       "units": "counter values are KiB per launch (average over launches); FETCH_SIZE x 2 on gfx950",
       "kernels": {}, "hbm_bytes_per_launch": {}}
for k in sorted(fetch):
    hbm = int((2.0 * fetch[k] + write.get(k, 0.0)) * 1024)
    out["kernels"][k] = {"launches": nf[k], "FETCH_SIZE_kib": round(fetch[k], 1),
                         "WRITE_SIZE_kib": round(write.get(k, 0.0), 1), "hbm_bytes_corrected": hbm}
    out["hbm_bytes_per_launch"][k] = hbm
grp = ["mn_cc_tiles", "mn_cc_borders", "mn_cc_flatten", "mn_cc_hook"]      # one launch each per image
if all(g in out["hbm_bytes_per_launch"] for g in grp):
    out["hbm_bytes_per_launch"]["mn_cc_tiles+mn_cc_borders+mn_cc_flatten+mn_cc_hook"] = sum(
        out["hbm_bytes_per_launch"][g] for g in grp)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["hbm_bytes_per_launch"], indent=1))
