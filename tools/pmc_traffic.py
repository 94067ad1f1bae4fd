"""HBM traffic of each kernel family from rocprofv3 PMC passes (separate passes for FETCH_SIZE and WRITE_SIZE, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: they do not fit one pass on gfx950).

    python tools/pmc_traffic.py [--reads 20000] [--ref-mb 3000]

Runs `bench.py --steps 1 --warmup 0 --no-cpu-baseline` with MM2AMD_LANES=1 under `rocprofv3 --pmc <counter> --kernel-trace`,
sums each counter per kernel name, divides by the launch count and writes profiles/pmc_traffic.json:
    { "<kernel family>": {"fetch_bytes_per_launch": .., "write_bytes_per_launch": .., "launches": .., "note": ..}, ... }
Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction from the guide: FETCH_SIZE counts 128-B requests
as 64 B for wide coalesced streaming reads, so the doubled value is given as well ("fetch_bytes_x2"); other access widths
and WRITE_SIZE are uncalibrated on this part -- treat them as lower bounds / ratios."""
import argparse
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name):
    n = name.split("(")[0].replace("void ", "").replace("mm2amd::", "")
    return n.split("<")[0] if n.startswith(("ksw_stream_kernel", "ksw_gapfill_kernel", "ksw_splice_kernel", "ksw_extd2_kernel", "ksw_band_kernel", "ksw_extq_kernel", "ksw_ext_kernel", "anchor_sort_kernel", "chain_fill_kernel", "sketch_wave_kernel")) else n


ALG = {}  # family -> [algorithmic bytes, launches] from the bench line of the last pass


def collect(counter, args):
    ALG.clear()
    out = os.path.join(ROOT, "gpurun_out", "pmc_" + counter)
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, MM2AMD_LANES="1", TMPDIR="/tmp", MM2AMD_DEVICE_FINISH="1")  # (region_finish_kernel on, so that it is counted too)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--reads", str(args.reads), "--ref-mb", str(args.ref_mb), "--preset", args.preset]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    try:  # the bench line of the same run: algorithmic bytes and launches per launch class, as the launches account them
        line = json.loads(p.stdout.decode().strip().split("\n")[-1])
        for k, b in line["roofline"]["kernels_alg_bytes"].items():
            f = family(k.split("[")[0])
            a = ALG.setdefault(f, [0.0, 0])
            a[0] += b
            a[1] += line["roofline"]["kernels_launches"][k]
    except Exception as e:  # noqa: BLE001
        print("no bench line from the %s pass: %s" % (counter, e), file=sys.stderr)
    db = sqlite3.connect(glob.glob(os.path.join(out, "*.db"))[0])
    cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
    res = {}
    for kn, cn, v, disp in db.execute("select %s, %s, sum(%s), count(distinct dispatch_id) from counters_collection group by %s, %s" % (name_col, cnt_col, val_col, name_col, cnt_col)):
        if cn != counter:
            continue
        f = family(kn)
        a = res.setdefault(f, [0.0, 0])
        a[0] += v
        a[1] += disp
    subprocess.run(["rm", "-rf", out])
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--ref-mb", type=float, default=3000)
    ap.add_argument("--out", default=None, help="where to write the JSON (default: profiles/pmc_traffic.json; on the GPU box give a path under gpurun_out/)")
    ap.add_argument("--preset", default="map-ont", help="map-ont (default) or splice; results of a run are merged into the existing JSON")
    a = ap.parse_args()
    fetch = collect("FETCH_SIZE", a)
    write = collect("WRITE_SIZE", a)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    # a map-ont run replaces every entry but the splice kernel's (which it does not exercise); a splice run only adds that one
    out = {k: v for k, v in old.items() if (k.startswith("ksw_splice") if a.preset == "map-ont" else True)}
    for f in sorted(set(fetch) | set(write)):
        if not any(f.startswith(p) for p in ("ksw_", "chain_", "seed_", "sketch", "anchor_", "encode", "region_", "rechain_", "first_chain", "gather_chains")):
            continue
        if a.preset != "map-ont" and not f.startswith("ksw_splice"):
            continue  # a splice run only contributes the kernel that the map-ont run does not exercise
        fb, fl = fetch.get(f, [0.0, 0])
        wb, wl = write.get(f, [0.0, 0])
        n = max(fl, wl, 1)
        alg = ALG.get(f) or ALG.get({"sketch_wave_kernel": "sketch_kernel", "anchor_sort_ties_kernel": "anchor_sort_kernel"}.get(f, f))
        alg_per = alg[0] / max(alg[1], 1) if alg else None
        # gfx950: FETCH_SIZE reports half the bytes of coalesced dword / qword / dwordx4 streams (x2 correction: profiles/r03_pmc_calibration.json) but a
        # whole 64-byte sector for a random 8-byte probe (x1): the index-probing kernel is priced without the correction
        ff = 1.0 if f in ("seed_collect_kernel",) else 2.0
        traffic = ff * fb * 1024 / n + wb * 1024 / n
        out[f] = {"fetch_bytes_per_launch": fb * 1024 / n, "fetch_bytes_x2": 2 * fb * 1024 / n, "write_bytes_per_launch": wb * 1024 / n, "launches": n,
                  "fetch_factor": ff, "traffic_bytes_per_launch": traffic,
                  "alg_bytes_per_launch": alg_per, "traffic_over_algorithmic": round(traffic / alg_per, 2) if alg_per else None,
                  "commit": os.environ.get("MM2AMD_COMMIT"),
                  "note": "%s: per launch at %d reads vs %d Mb; FETCH_SIZE/WRITE_SIZE in KiB -> bytes; x2 = gfx950 wide-read correction; WRITE_SIZE uncalibrated" % (a.preset, a.reads, a.ref_mb)}
    json.dump(out, open(a.out or path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
