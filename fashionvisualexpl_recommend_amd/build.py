"""In-tree build of libbprx.so (hipcc, gfx950 only).  The .so is git-ignored but travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libbprx.so")
SOURCES = ["bprx_api.hip", "bprx_sparse.hip", "bprx_proj.hip", "bprx_philox.hip", "bprx_eval.hip", "bprx_probe.hip", "bprx_route.hip", "bprx_sampler.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "bprx.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into fashionvisualexpl_recommend_amd/libbprx.so."""
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.rsplit(".", 1)[0] + ".o")
        cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", INCLUDE, "-I", CSRC,
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
