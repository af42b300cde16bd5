"""Build libsuhpe.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m semiuhpe_amd.build          # or semiuhpe_amd.build.build()

hipcc cross-compiles without a GPU; the .so stays next to the sources (semiuhpe_amd/csrc/libsuhpe.so) so that it
travels with the repo snapshot to the GPU box.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsuhpe.so")
SOURCES = ["common.hip", "head.hip", "nn_elem.hip", "layout_aug.hip", "conv_igemm.hip", "conv_wino.hip", "conv_wino_small.hip", "dwconv.hip", "attention.hip", "se_fused.hip", "augment.hip", "comm.hip", "raster.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inl"))]
    hs.append(os.path.join(HERE, "..", "include", "suhpe.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = _newest_header()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, source, defines):
    """libsuhpe_<name>.so = the library with `source` recompiled under extra -D flags (timing experiments: SUHPE_LIB=<path> selects it)"""
    build(verbose=False)
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    obj = os.path.join(objdir, source.replace(".hip", f".{name}.o"))
    subprocess.check_call([hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, source), "-o", obj])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) if s != source else obj for s in SOURCES]
    out = os.path.join(CSRC, f"libsuhpe_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
