"""Compile the HIP sources in csrc/ into zest-nerf_amd/libzest_hip.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels with the tree to the GPU box.  `python build_hip.py [-f]`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libzest_hip.so")
SOURCES = ["capi.hip", "composite.hip", "composite_bwd.hip", "encode.hip", "rays.hip", "volume_cost.hip", "losses.hip", "mlp_plan.hip", "mlp.hip", "mlp_bf16.hip", "mlp_train.hip",
           "fused.hip", "fused_s0.hip", "fused_s2.hip", "fused_s4.hip", "fused_s0d0.hip",
           "fused_s4d2.hip", "fused_s2d2.hip", "fused_s2d0.hip", "fused_s4d0.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-inline-asm",     # the LDS-DMA asm declares the reserved register m0 clobbered on purpose
         # the network is one fully unrolled body per kernel: keep `#pragma unroll` honoured
         "-mllvm", "-pragma-unroll-threshold=1000000"]


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    hdr.append(os.path.join(HERE, "..", "include", "zest_render.h"))
    return hdr


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def _compile(src, extra):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _deps()):
        cmd = [HIPCC] + FLAGS + extra + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, extra_flags=(), tag=None, only=None):
    """tag: build an experiment variant libzest_hip_<tag>.so with extra -D flags (objects kept
    apart); only: restrict to these sources (others are taken from the default build)."""
    global OBJ, LIB
    if tag:
        OBJ = os.path.join(CSRC, "build_" + tag)
        LIB = os.path.join(HERE, "libzest_hip_%s.so" % tag)
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s)) and (not only or s in only)]
    if force:
        for s in srcs:
            o = os.path.join(OBJ, s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=min(7, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, list(extra_flags)), srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lrocblas"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    # python build_hip.py [-f] [--tag NAME -DFOO=1 ...]
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else None
    print(build(force="-f" in argv, extra_flags=[a for a in argv if a.startswith("-D")], tag=tag))
