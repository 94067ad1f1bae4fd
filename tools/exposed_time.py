"""Which kernels are on the critical path?  From a rocprofv3 rocpd database (--kernel-trace): over the last T seconds of kernel
activity, split the wall clock by WHAT was running:
  * idle                  no kernel at all
  * only latency-bound    nothing but kernels of the 'latency' families below (few busy CUs) -- charged to the family running
  * throughput            at least one throughput-bound kernel (the streaming / gap-fill / seeding kernels) was running
and, for every family, the time it ran with no kernel of another family beside it.
    python tools/exposed_time.py results.db [window_seconds]
Measurement scaffolding."""
import sqlite3
import sys

LATENCY = ("ksw_extd2_kernel", "region_finish_kernel", "chain_rmq_kernel")  # plus ksw_ext_kernel with 8 register sets (see fam())


def fam(name):
    n = name.split("(")[0]
    base = n.split("<")[0].split(" ")[-1].replace("mm2amd::", "")
    if base in ("ksw_ext_kernel", "ksw_extq_kernel"):
        return base + "<..,8>" if ", 8>" in n or ",8>" in n else base + "<..,4>"
    if base == "ksw_gapfill_kernel":
        return "ksw_gapfill_kernel"
    return base


def is_latency(f):
    return f in LATENCY or f in ("ksw_ext_kernel<..,8>", "ksw_extq_kernel<..,8>")


def main():
    db = sqlite3.connect(sys.argv[1])
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    rows = db.execute("select start, end, name from kernels order by start").fetchall()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(win * 1e9)
    ev = []
    for s, e, nm in rows:
        if e <= t0:
            continue
        f = fam(nm)
        ev.append((max(s, t0), 1, f))
        ev.append((e, -1, f))
    ev.sort()
    live = {}
    last = t0
    idle = 0
    thr = 0
    lat_only = {}
    alone = {}
    conc_sum = 0
    for t, d, f in ev:
        dt = t - last
        if dt > 0:
            act = [k for k, v in live.items() if v > 0]
            conc_sum += dt * sum(live.values())
            if not act:
                idle += dt
            elif all(is_latency(k) for k in act):
                key = "+".join(sorted(act))
                lat_only[key] = lat_only.get(key, 0) + dt
            else:
                thr += dt
            if len(act) == 1:
                alone[act[0]] = alone.get(act[0], 0) + dt
        live[f] = live.get(f, 0) + d
        last = t
    W = win * 1e9
    print("window %.2f s: idle %.1f %%, only latency-bound kernels running %.1f %%, a throughput-bound kernel running %.1f %%; mean kernels in flight %.2f"
          % (win, 100 * idle / W, 100 * sum(lat_only.values()) / W, 100 * thr / W, conc_sum / W))
    for k, v in sorted(lat_only.items(), key=lambda kv: -kv[1])[:8]:
        print("  only %-60s %7.1f ms per s" % (k, v / W * 1e3))
    print("time a family ran with no other family beside it (ms per s of wall):")
    for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:16]:
        print("  %-44s %7.1f" % (k, v / W * 1e3))


if __name__ == "__main__":
    main()
