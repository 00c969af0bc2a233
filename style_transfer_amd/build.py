"""Builds libst_kernels.so (hipcc, gfx950 only) in-tree, next to this file.  Every source is compiled to its own object
(in parallel, only when it or a header changed), then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libst_kernels.so")
SOURCES = ["st_decode.hip", "st_scorer.hip", "st_accept.hip", "st_simloss.hip", "st_encode.hip", "st_fgim.hip", "st_train.hip", "st_aetrain.hip", "st_logfmt.cpp"]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "st_kernels.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    t_hdr = max(os.path.getmtime(h) for h in headers())
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", obj, src])
    workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "8"))))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s) + ".o") for s in sources()]
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
