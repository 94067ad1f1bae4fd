"""FETCH_SIZE / WRITE_SIZE calibration factors for our access patterns (tools/pmc_calib.hip): reported bytes / true bytes per pattern.
    python tools/pmc_calib.py > gpurun_out/r03_pmc_calibration.json      (on the GPU box; two rocprofv3 --pmc passes)"""
import glob, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "build", "pmc_calib")
truth = {}
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out = os.path.join(ROOT, "gpurun_out", "pmc_calib_" + counter)
    subprocess.run(["rm", "-rf", out])
    p = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out, "-o", "pmc", "--", exe], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    for line in p.stdout.decode().split("\n"):
        if line.startswith("CALIB"):
            _, k, w, r = line.split()
            truth[k] = (int(w), int(r))
    db = sqlite3.connect(glob.glob(os.path.join(out, "**", "*.db"), recursive=True)[0])
    cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
    for kn, cn, v in db.execute("select %s, %s, sum(%s) from counters_collection group by %s, %s" % (name_col, cnt_col, val_col, name_col, cnt_col)):
        k = kn.split("(")[0].replace("void ", "")
        if k in truth and cn == counter:
            res.setdefault(k, {})[counter + "_KiB"] = v
    subprocess.run(["rm", "-rf", out])
for k, (w, r) in truth.items():
    e = res.setdefault(k, {})
    e["true_write_bytes"], e["true_read_bytes"] = w, r
    if w:
        e["write_reported_over_true"] = e.get("WRITE_SIZE_KiB", 0) * 1024 / w
    if r:
        e["fetch_reported_over_true"] = e.get("FETCH_SIZE_KiB", 0) * 1024 / r
    if k == "ld_random8":
        e["fetched_bytes_per_probe_reported"] = e.get("FETCH_SIZE_KiB", 0) * 1024 / (r / 8)
print(json.dumps({"cmd": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- tools/build/pmc_calib", "note": "reported = counter in KiB x 1024; factors are what tools/pmc_traffic.py divides by for the matching access pattern", "kernels": res}, indent=1, sort_keys=True))
