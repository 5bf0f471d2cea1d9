"""Per-kernel MFMA roofline of the CNN workloads from the round's rocprofv3 summaries: average duration (kernel_stats_cnn*.csv), arithmetic per launch
(1024 frames), achieved TFLOP/s against the dense fp32 MFMA peak (157.3, MI355X_MICROARCH.md), and the counter-based MFMA utilisation beside it.
usage: python tools/cnn_roofline.py profiles/r02 > profiles/r02_cnn_kernel_roofline.json"""
import csv, json, sys
pre = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02"
B, PEAK = 1024, 157.3
FLOP = {      # per frame
    "64": {"k_conv12": 2 * 60 * 60 * 25 * 16 + 2 * 12 * 12 * 256 * 64, "k_conv1<": 2 * 60 * 60 * 25 * 16, "k_conv2": 2 * 12 * 12 * 256 * 64, "k_fc<true, 2, 4, true": 2 * 2304 * 2048, "k_fc144": 2 * 2048 * 2304},
    "128": {"k_conv1<": 2 * 124 * 124 * 25 * 16, "k_conv2": 2 * 28 * 28 * 256 * 64, "k_fc<true, 2, 4, true": 2 * 12544 * 2048, "k_fc144": 2 * 2048 * 2304},
}
out = {"_note": "frac = achieved / 157.3 TFLOP/s (dense fp32 MFMA); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs) from the separate counter pass; B = 1024 frames per launch"}
for side, stats, pmc in (("64", pre + "_rocprofv3_kernel_stats_cnn.csv", pre + "_pmc_mfma_util.json"), ("128", pre + "_rocprofv3_kernel_stats_cnn128.csv", pre + "_pmc_mfma128_util.json")):
    util = json.load(open(pmc))
    rows = list(csv.DictReader(open(stats)))
    tot_us = tot_fl = 0.0
    tab = {}
    for r in rows:
        name = r["Name"]
        key = next((k for k in FLOP[side] if name.replace("void ", "").startswith(k.split("<")[0]) and (("<" not in k) or k in name)), None)
        us = float(r["AverageNs"]) / 1e3
        if key is None:
            if "softmax" in name: tab["k_softmax_decode"] = {"avg_us": round(us, 1)}; tot_us += us
            continue
        fl = FLOP[side][key] * B
        u = next((v["mfma_util"] for k, v in util.items() if isinstance(v, dict) and k.startswith(key.split("<")[0]) and (("<" not in key) or key in k)), None)
        tab[name.split("(")[0].replace("void ", "")] = {"avg_us": round(us, 1), "gflop_per_launch": round(fl / 1e9, 2), "tflops": round(fl / us / 1e6, 1), "frac": round(fl / us / 1e6 / PEAK, 3), "mfma_util": u}
        tot_us += us; tot_fl += fl
    tab["_all_kernels"] = {"sum_us": round(tot_us, 1), "tflops": round(tot_fl / tot_us / 1e6, 1), "frac": round(tot_fl / tot_us / 1e6 / PEAK, 3)}
    out["input_%sx%s" % (side, side)] = tab
print(json.dumps(out, indent=1))
