"""Builds artspeech_amd/lib/libartspeech_hip.so (gfx950 only) with hipcc.

In-tree on purpose: the .so travels to the GPU box with the repo snapshot.  Sources that are older
than their object are not recompiled.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libartspeech_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# AS_BUILD_FLAGS="-DAS_EXPERIMENTS": also compile the tuning variants (other GEMM pipeline shapes, the 256x128 tile) that the
# shipped library leaves out; AS_TEST_EXPERIMENTS=1 adds their ids to the test matrix
EXTRA = os.environ.get("AS_BUILD_FLAGS", "").split()
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")] + EXTRA


def _compile(src):
    obj = os.path.join(LIBDIR, "obj", os.path.basename(src) + ".o")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    if os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
        return obj, False
    subprocess.check_call([HIPCC] + FLAGS + ["-c", src, "-o", obj])
    return obj, True


def build_lib(verbose=True):
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    stamp = os.path.join(LIBDIR, "obj", "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(EXTRA):          # other flags: every object is stale
        for o in glob.glob(os.path.join(LIBDIR, "obj", "*.o")):
            os.remove(o)
        open(stamp, "w").write(" ".join(EXTRA))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_lib()
