import os
"""SQ counters per kernel family (one rocprofv3 --pmc pass; counters given on the command line).
    python tools/pmc_sq.py SQ_WAVE_CYCLES SQ_INSTS_VALU ... [--reads N]"""
import glob, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reads = "20000"
if "--reads" in sys.argv:
    reads = sys.argv[sys.argv.index("--reads") + 1]
    args = [a for a in args if a != reads]
out = os.path.join(ROOT, "gpurun_out", "pmc_sq")
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, MM2AMD_LANES="1", TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--pmc"] + args + ["--kernel-trace", "-d", out, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                "--no-cpu-baseline", "--reads", reads], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
db = sqlite3.connect(glob.glob(os.path.join(out, "*.db"))[0])
cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
name_col = "kernel_name" if "kernel_name" in cols else "name"
val_col = "value" if "value" in cols else "counter_value"
cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
res = {}
for kn, cn, v in db.execute("select %s, %s, sum(%s) from counters_collection group by %s, %s" % (name_col, cnt_col, val_col, name_col, cnt_col)):
    k = kn.split("(")[0].replace("void ", "").replace("mm2amd::", "")
    if not any(k.startswith(p) for p in ("ksw_", "chain_", "seed_", "sketch", "anchor_")):
        continue
    res.setdefault(k, {})[cn] = v
for k in sorted(res):
    print(k, " ".join("%s=%.6g" % (c, res[k].get(c, 0)) for c in args))
import json
json.dump({"commit": os.environ.get("MM2AMD_COMMIT"), "cmd": "MM2AMD_LANES=1 rocprofv3 --pmc %s --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --reads %s" % (" ".join(args), reads),
           "note": "sums over all dispatches of a kernel; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles per wave (MI355X_MICROARCH.md)", "kernels": res},
          open(os.path.join(ROOT, "gpurun_out", "pmc_sq_%s.json" % os.environ.get("PMC_SQ_TAG", "last")), "w"), indent=1, sort_keys=True)
subprocess.run(["rm", "-rf", out])
