"""VALU instructions per register-set row of the gap-fill DP kernels' hot loops, counted in the gfx950 assembly of the working tree (hipcc -S; no GPU needed).

    python tools/isa_row_counts.py            -> profiles/isa_row_counts.json

bench.py's `roofline.valu` prices the streaming kernel's row with these counts instead of constants typed into it.  A kernel's
hot loop is found the way tools/isa_scratch_report.sh finds it: the line ranges that hold the packed 16-bit arithmetic of the DP cell (v_pk_*); the
unrolled row body appears twice per kernel (the two halves of the double-buffered row loop), each copy covers the kernel's NC register sets, so one
register-set row = a range's count / NC, averaged over the two copies.  Classes by encoding (what the issue table profiles/r03_valu_issue_bench_v1.txt
distinguishes): VOP3P (v_pk_*), DPP (…_dpp), SDWA (…_sdwa), VOP3 (…_e64 and the three-operand forms that only exist there), VOP2/VOP1/VOPC (the rest).
The JSON carries the commit and the SHA-256 of the sources it was made from; bench.py reports whether they are the sources it runs."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minimap2_amd", "csrc")
SOURCES = ["ksw_stream.hip", "ksw_gapfill.hip", "ksw_band.hip", "ksw_band.hpp", "ksw_gapfill_dev.hpp", "ksw_pk.hpp", "ksw_dev.hpp"]
VOP3_ONLY = ("v_perm_b32", "v_add3_u32", "v_and_or_b32", "v_bfi_b32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_or3_b32", "v_xad_u32", "v_mad_", "v_med3_", "v_min3_", "v_max3_",
             "v_alignbit_b32", "v_alignbyte_b32", "v_bfe_", "v_mul_lo_u32", "v_mul_hi_u32", "v_readlane", "v_writelane", "v_cndmask_b32_e64", "v_lshlrev_b64", "v_lshrrev_b64", "v_mad_u64_u32")


def source_digest():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def classify(ins):
    op = ins.split()[0]
    if not op.startswith("v_"):
        return None
    if op.startswith("v_pk_"):
        return "vop3p"
    if "_dpp" in op or " row_" in ins or "quad_perm" in ins or "wave_" in ins and "dpp" in ins:
        return "dpp"
    if "_sdwa" in op:
        return "sdwa"
    if op.endswith("_e64") or op.startswith(VOP3_ONLY):
        return "vop3"
    return "vop2"


def kernels_of(asm):
    lines = asm.split("\n")
    names = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
    for i, n in names:
        e = next((x for x in ends if x > i), len(lines))
        yield n, lines[i:e]


def hot_ranges(body):
    pk = [i for i, l in enumerate(body) if "\tv_pk_" in l]
    rng = []
    for i in pk:
        if rng and i - rng[-1][1] < 60:
            rng[-1][1] = i
        else:
            rng.append([i, i])
    return [r for r in rng if sum(1 for i in pk if r[0] <= i <= r[1]) >= 30]


def main():
    out = {}
    for f in ("ksw_stream.hip", "ksw_gapfill.hip", "ksw_band.hip"):
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-x", "hip",
                                   "--cuda-device-only", "-S", os.path.join(CSRC, f), "-o", tmp.name], stderr=subprocess.DEVNULL)
            asm = open(tmp.name).read()
        for name, body in kernels_of(asm):
            dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
            m = re.search(r"(ksw_\w+_kernel)<(\d+), *(\d+)(?:, *\d+)?>", dem)  # (the banded kernel has a third argument: the window size its LDS arrays are laid out for)
            if not m:
                continue
            # register sets per row: the streaming kernel's first template argument; the strip kernel always sweeps four sets of 64 columns
            # (the banded kernel: its first template argument too; a copy of its body is one row -- even or odd -- of the row pair)
            n_sets = int(m.group(2)) if m.group(1) in ("ksw_stream_kernel", "ksw_band_kernel") else 4
            copies = []
            ranges = hot_ranges(body)
            if m.group(1) == "ksw_band_kernel":
                # the banded kernel's row-pair loop as the compiler laid it out: the inner loop (its header's label up to the last branch back to it) that holds the
                # keyed cells -- DPP moves, the LDS loads' address arithmetic and the loop's own bookkeeping included
                # (the loop may be rotated, its body laid out before its header: take the basic blocks the assembler's comments assign to a loop header)
                blocks, cur = {}, None
                for i, l in enumerate(body):
                    mm_ = re.match(r"^\.LBB(\d+)_(\d+):(.*)", l)
                    if mm_:
                        hdr = re.search(r"Header=BB\d+_(\d+)", mm_.group(3))
                        cur = hdr.group(1) if hdr else (mm_.group(2) if i + 1 < len(body) and "Inner Loop Header" in (mm_.group(3) + body[i + 1]) else None)
                    if cur is not None:
                        blocks.setdefault(cur, []).append(i)
                loops = [v for v in blocks.values() if sum(1 for i in v if "\tv_pk_" in body[i]) >= 60]
                loops.sort(key=len)
                band_lines = loops[0] if loops else []
                ranges = [[0, -1]] if band_lines else []
            for a, b in ranges:
                c = {"vop3p": 0, "dpp": 0, "sdwa": 0, "vop3": 0, "vop2": 0}
                for l in ([body[i] for i in band_lines] if m.group(1) == "ksw_band_kernel" else body[a:b + 1]):
                    l = l.strip()
                    if l.startswith("v_"):
                        k = classify(l)
                        if k:
                            c[k] += 1
                copies.append(c)
            if not copies:
                continue
            set_rows = len(copies) * n_sets
            if m.group(1) == "ksw_band_kernel":
                # the compiler peels and merges the banded kernel's even and odd rows as it likes: a range's register-set rows = its keyed cells (34 packed operations each)
                set_rows = sum(round(c["vop3p"] / 34.0) for c in copies)
            per = {k: round(sum(c[k] for c in copies) / max(set_rows, 1), 2) for k in copies[0]}
            per["total"] = round(sum(per.values()), 2)
            out["%s<%s,%s>" % m.groups()] = {"register_sets": n_sets, "row_copies": len(copies), "per_register_set_row": per, "per_copy": copies}
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE).stdout.decode().strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--"] + [os.path.join("minimap2_amd", "csrc", f) for f in SOURCES], stdout=subprocess.PIPE).stdout.strip())
    doc = {"made_by": "tools/isa_row_counts.py (hipcc -S --offload-arch=gfx950 -O3, the build's flags)", "commit": commit + ("+uncommitted changes to the sources" if dirty else ""),
           "sources": SOURCES, "sources_sha256": source_digest(), "kernels": out,
           "lane_utilisation": {"ksw_stream_kernel": 0.872, "ksw_gapfill_kernel": 0.727, "ksw_band_kernel": 1.0,
                                "basis": "cells / (128 x executed register-set rows) of the MM2AMD_GF_COUNT build: profiles/r04_stream_lane_utilisation_emu.txt (streaming kernel, HEAD's schedule on the wave emulator: a property of the schedule and the job mix), profiles/r02_stream_lane_utilisation.txt (MI355X)"}}
    path = os.path.join(ROOT, "profiles", "isa_row_counts.json")
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print(k, v["per_register_set_row"])
    print("->", path)


if __name__ == "__main__":
    main()
