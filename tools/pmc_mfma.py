"""MFMA utilisation of the CNN kernels from one rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE).

MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
kernel's cycles = GRBM_GUI_ACTIVE / 8 and utilisation = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs)."""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
out = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE on `bench.py --workload cnn` (B=1024); means per dispatch; "
                "mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)"}
for k, v in acc.items():
    n = max(1, cnt[k])
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    if mf <= 0:
        continue
    out[k] = {"dispatches": n, "kernel_cycles": int(gui), "mfma_busy_cycles": int(mf), "sq_busy_cu_cycles": int(v.get("SQ_BUSY_CU_CYCLES", 0.0) / n),
              "mfma_util": round(mf / (gui * 1024.0), 4) if gui > 0 else None}
print(json.dumps(out, indent=1))
