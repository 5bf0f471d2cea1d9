# Reads a rocprofv3 kernel trace CSV and reports, per queue, busy time and how much kernels of different queues overlap in time.
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:40]) for r in rows))
t0 = ev[0][0]
byq = collections.Counter(); 
for s, e, q, n in ev: byq[q] += e - s
span = max(e for s, e, q, n in ev) - t0
print("kernels", len(ev), "span ms", span / 1e6, "queues", {q: round(v / 1e6, 2) for q, v in byq.items()})
# union busy time
pts = sorted([(s, 1) for s, e, q, n in ev] + [(e, -1) for s, e, q, n in ev])
depth = 0; last = None; hist = collections.Counter()
for t, d in pts:
    if last is not None: hist[depth] += t - last
    depth += d; last = t
print("time by number of kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
if len(sys.argv) == 3:
    for s, e, q, n in ev[-int(sys.argv[2]):]: print("%10.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
if len(sys.argv) > 3:      # window: first kernel index, count
    a, n = int(sys.argv[2]), int(sys.argv[3])
    for s, e, q, nm in ev[a:a + n]: print("%10.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm))
