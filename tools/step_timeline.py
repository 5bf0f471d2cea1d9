# Prints the kernel timeline of the LAST step of a rocprofv3 --kernel-trace CSV of bench.py (one line per kernel: start us, duration us, gap to the
# previous kernel's end on ANY queue, queue, name) and a summary of the dependent gaps.   usage: python tools/step_timeline.py trace.csv [kernels_per_step]
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:44]) for r in rows))
# a step starts with k_prepare
starts = [i for i, e in enumerate(ev) if e[3].startswith("k_prepare")]
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0 = step[0][0]
busy_end = t0
gaps = []
for s, e, q, n in step:
    gap = (s - busy_end) / 1e3
    print("%9.1f %8.1f %7.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n))
    if s > busy_end: gaps.append(gap)
    busy_end = max(busy_end, e)
print("kernels in the step:", len(step), "span ms: %.3f" % ((busy_end - t0) / 1e6), "idle (no kernel running) ms: %.3f in %d gaps, median gap us %.1f" % (sum(gaps) / 1e3, len(gaps), sorted(gaps)[len(gaps) // 2] if gaps else 0))
by = collections.Counter()
for s, e, q, n in step: by[n] += (e - s) / 1e3
print({k: round(v, 1) for k, v in by.most_common()})
