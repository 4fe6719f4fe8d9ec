"""Builds libafhip.so (gfx950) in-tree with hipcc.  No GPU is needed to build.

    python -m af_mi355x.build        (or __graft_entry__.build())
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libafhip.so")
SOURCES = ["af_api.hip", "af_pack.hip", "af_pool.hip", "af_stem.hip", "af_stem_pool.hip", "af_stem3.hip", "af_conv.hip", "af_conv133.hip", "af_conv133g.hip", "af_conv311.hip", "af_conv_ca.hip", "af_conv_cpa.hip", "af_conv111.hip", "af_ftcn.hip", "af_dual.hip", "af_conv_small.hip", "af_align.hip", "af_block_abc.hip"]
HEADERS = [os.path.join(CSRC, "af_common.h"), os.path.join(os.path.dirname(HERE), "include", "af_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
