"""Builds rlvi_amd/librlvi_gfx950.so (hipcc, gfx950 only) in-tree.

One object per .hip (compiled in parallel into rlvi_amd/csrc/build/), then one link."""
import glob
import os
import shlex
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "librlvi_gfx950.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale():
    return _newer(LIB, sources() + headers())


def build(force=False, verbose=False, extra_flags=None, lib=LIB):
    extra = list(extra_flags or shlex.split(os.environ.get("RLVI_EXTRA_FLAGS", "")))
    # (variant builds -- extra flags, tools/build_variants.py -- get an object directory named after their flags:
    #  a stable digest, so that rebuilding a variant reuses its directory instead of leaving a new one behind)
    import hashlib
    objdir = OBJ if lib == LIB and not extra else OBJ + "_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]
    os.makedirs(objdir, exist_ok=True)
    hdrs = headers()
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj])
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in sources()]
    # an object whose source is gone (a kernel file that was removed or renamed) must not linger
    for old in glob.glob(os.path.join(objdir, "*.o")):
        if old not in objs:
            os.remove(old)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
