"""HBM traffic per launch of this repo's kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for gfx950: the counters are in KB; FETCH_SIZE reports half of
the bytes of 16-B-per-lane streaming reads (global_load and buffer_load ... lds alike) -> doubled; WRITE_SIZE is taken
as reported (uncalibrated on gfx950).  Usage: traffic_from_pmc.py <fetch_dir> <write_dir>  -> JSON on stdout."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items() if re.match(r"^(void )?k_", k)}


fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"units": "bytes per launch", "correction": "FETCH_SIZE [KB] x 1024 x 2 (gfx950 wide-read under-count), WRITE_SIZE [KB] x 1024", "kernels": {}}
for k in fe:
    out["kernels"][k.split("(")[0].replace("void ", "")] = {
        "launches": fe[k][1], "fetch_bytes": fe[k][0] * 1024 * 2, "write_bytes": wr.get(k, (0, 0))[0] * 1024,
        "hbm_bytes": fe[k][0] * 1024 * 2 + wr.get(k, (0, 0))[0] * 1024}
print(json.dumps(out, indent=1))
