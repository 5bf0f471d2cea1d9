"""Aggregates two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per kernel launch.

MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Other access widths are uncalibrated."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as fp:
        for r in csv.DictReader(fp):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


# usage: pmc_traffic.py fetch.csv write.csv [workload [frames_per_gpu]]   (what the two passes were run on; bench.py only quotes a file for the workload it names)
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
workload = sys.argv[3] if len(sys.argv) > 3 else "cnn+solver"
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate runs of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` on the workload below, "
                "mean per dispatch, KB. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads, so "
                "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; other access widths are uncalibrated.",
       "_measured_on": {"workload": workload, "frames_per_gpu": frames}}
for k in fetch:
    if k.startswith('_'):
        continue
    f = fetch[k][0] / max(1, fetch[k][1])
    w = write[k][0] / max(1, write[k][1]) if k in write else 0.0
    out[k] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "dispatches": fetch[k][1], "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
print(json.dumps(out, indent=1))
