"""Build libmm2amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m minimap2_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the
GPU box with the repository snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmm2amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, force, hdr_t):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
        return obj
    if src.endswith(".hip"):
        cmd = [HIPCC] + COMMON + ["-x", "hip", "-c", path, "-o", obj]
    else:  # host-only translation units: the same compiler driver, host pass only
        cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
               "-I" + os.path.join(os.path.dirname(HERE), "include"), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _newest_header()
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_t), _sources()))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
