"""Turn the rocprofv3 --pmc passes of scripts/collect_traffic.sh into profiles/traffic_latest.json."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
N = 512 ** 3
KIB = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KiB


def per_kernel(mode, ctr):
    f = glob.glob(os.path.join(out, f"{mode}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"rbgs3_fused_k(<[^>]*>)", r["Kernel_Name"])
        if r["Counter_Name"] != ctr or not m:
            continue
        acc.setdefault(m.group(1), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


import subprocess, time
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = "unknown (no .git on the GPU box)"
res = {"collected": time.strftime("%Y-%m-%d") + ", rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/collect_traffic.sh",
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/collect_traffic.sh): level-1 fused "
               "smoother launches at 512^3 (7 sweeps = 2+2+2+1), per-launch averages in bytes. FETCH_SIZE is doubled as "
               "MI355X_MICROARCH.md prescribes (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B). "
               "Kernel template arguments: <scalar type, sweeps per launch, tile x, tile y, threads, waves/SIMD, "
               "zero-rhs variant, mode (0 plain, 1 +residual, 2 +metric, 3 +prolongation)>",
       "kernels": {}}
for mode in ("general", "zero"):
    fe, cnt = per_kernel(mode, "FETCH_SIZE")
    wr, _ = per_kernel(mode, "WRITE_SIZE")
    for k in fe:
        sweeps = int(k.strip("<>").split(",")[1])   # <type, sweeps, tile x, ...>
        res["kernels"][f"{mode} {k}"] = {
            "launches_sampled": cnt[k], "sweeps_per_launch": sweeps,
            "fetch_bytes_corrected": 2 * fe[k] * KIB, "write_bytes": wr.get(k, 0.0) * KIB,
            "total_bytes": 2 * fe[k] * KIB + wr.get(k, 0.0) * KIB,
            "algorithmic_bytes_per_launch": sweeps * N * (24 if mode == "general" else 16)}
gen2 = [v for k, v in res["kernels"].items() if k.startswith("general <double, 2")]
if gen2:  # the launch bench.py's roofline line is quoted on
    res["smoother_sweep_bytes_per_launch"] = gen2[0]["total_bytes"]
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
