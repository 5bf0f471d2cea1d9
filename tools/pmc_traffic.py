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


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate runs of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` (B=1024), "
                "mean per dispatch, KB. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads, so "
                "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; other access widths are uncalibrated."}
for k in fetch:
    f = fetch[k][0] / max(1, fetch[k][1])
    w = write[k][0] / max(1, write[k][1]) if k in write else 0.0
    out[k] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "dispatches": fetch[k][1], "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
print(json.dumps(out, indent=1))
