"""Aggregates rocprofv3 counter passes on `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io` into per-kernel issue / wait / L2 figures (profiles/rNN_pmc_solve_issue.json).

    python tools/pmc_solve.py <dir with the passes' *_counter_collection.csv and one *_kernel_trace.csv> [kernel name prefixes ...]

Per kernel and dispatch (means): the raw counters, and
  clocks_per_instruction   SQ_WAVE_CYCLES / (VALU + SALU + LDS + SMEM + VMEM instructions): what one wave pays per issued instruction while it is resident
                           (k_solve runs one wave per SIMD at 1024 frames: the wave's own issue rate)
  valu_issue_frac          SQ_INSTS_VALU x 4 / SQ_WAVE_CYCLES: the share of a resident wave's cycles in which its SIMD's VALU port took one of its instructions
                           (a VALU instruction occupies the port 4 cycles; with one wave per SIMD = VALU instructions issued / SIMD cycles available)
  wait_any_frac / wait_inst_any_frac / active_inst_any_frac   of SQ_WAVE_CYCLES (MI355X_MICROARCH.md: disjoint, together ~ all wave cycles)
  l2_read_gbps             TCP_TCC_READ_REQ_sum x 64 B / the kernel's duration;  l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
SQ counters are sums over the chip's shader engines; SQ_WAVE_CYCLES and SQ_BUSY_CYCLES count in units of 4 clocks on this family (the quotients above take that into
account where a clock count is meant)."""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2:] or ["k_solve", "k_contacts_coop", "k_cloud_rows", "k_fit_error", "k_chamber"]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fp:
        for r in csv.DictReader(fp):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
dur = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as fp:
        for r in csv.DictReader(fp):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[name][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[name][1] += 1
out = {"_note": __doc__.split("\n\n")[2].strip(), "_measured_on": {"workload": "cnn+solver", "frames_per_gpu": int(os.environ.get("FRAMES", "1024")), "command": "rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"}}
for name in sorted(acc):
    if not any(name.startswith(w) for w in want):
        continue
    c = {k: v[0] / max(1, v[1]) for k, v in acc[name].items()}
    e = {"dispatches": max(v[1] for v in acc[name].values()), "counters_mean_per_dispatch": {k: round(v, 1) for k, v in sorted(c.items())}}
    if name in dur and dur[name][1]:
        e["duration_us_under_the_counters"] = round(dur[name][0] / dur[name][1] / 1e3, 1)
    insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    wc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0
    if wc and insts:
        e["clocks_per_instruction"] = round(wc / insts, 2)
        e["valu_issue_frac"] = round(c.get("SQ_INSTS_VALU", 0.0) * 4.0 / wc, 4)
        e["issue_frac"] = e["valu_issue_frac"]
        for k, n in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_frac")):
            if k in c:
                e[n] = round(c[k] * 4.0 / wc, 4)
        if "SQ_WAVES" in c and c["SQ_WAVES"]:
            e["instructions_per_wave"] = round(insts / c["SQ_WAVES"], 0); e["clocks_per_wave"] = round(wc / c["SQ_WAVES"], 0)
    if "TCP_TCC_READ_REQ_sum" in c and name in dur and dur[name][1]:
        e["l2_read_gbps"] = round(c["TCP_TCC_READ_REQ_sum"] * 64.0 / (dur[name][0] / dur[name][1]), 1)
        e["l2_gbps"] = e["l2_read_gbps"]
    if c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0) > 0:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    out[name] = e
print(json.dumps(out, indent=1))
