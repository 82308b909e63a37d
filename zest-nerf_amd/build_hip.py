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
SOURCES = ["capi.hip", "composite.hip", "composite_bwd.hip", "encode.hip", "rays.hip", "volume_cost.hip", "costreg.hip", "losses.hip", "mlp_plan.hip", "mlp.hip", "mlp_engine.hip", "mlp_train.hip", "mlp_train16.hip",
           "mlp_train16_dw.hip", "fused.hip"]
# per-source flags.  The engine kernels outside the fused renderer (standalone MLP, training forward, backward data and
# finishing kernels) are built without the SLP vectorizer for the reason given at VARIANTS below (training forward
# -5 %, data kernel -3 %, finishing -2 %); the weight-gradient kernel, a translation unit of its own, keeps it: it runs
# twice as long without (profiles/r03_ab_train_noslp.txt)
SOURCE_FLAGS = {"mlp_engine.hip": ["-fno-slp-vectorize"], "mlp_train16.hip": ["-fno-slp-vectorize"]}
# fused renderer instantiations: fused_variant.hip once per (operand type, feature shape);
# heaviest first so the pool drains evenly
_SHAPES = [("s4d2", 4, "true", 2, "false"), ("s2d2", 2, "true", 2, "false"), ("s4d0", 4, "true", 0, "false"),
           ("s2d0", 2, "true", 0, "false"), ("s0d0", 0, "true", 0, "false"), ("s4", 4, "false", 0, "false"),
           ("s2", 2, "false", 0, "false"), ("s0", 0, "false", 0, "false"),
           ("s4v", 4, "false", 0, "true"), ("s2v", 2, "false", 0, "true")]     # 'v2' static nets
_PRECS = [("x3", "ZEST_PREC_F16X3"), ("bf16", "ZEST_PREC_BF16"), ("f16", "ZEST_PREC_F16")]
VARIANTS = [("fused_%s_%s" % (pt, tag),
             ["-DZEST_V_PTAG=%s" % pt, "-DZEST_V_EP=%s" % ep, "-DZEST_V_TAG=%s" % tag, "-DZEST_V_NTS=%d" % nts,
              "-DZEST_V_DYN=%s" % dyn, "-DZEST_V_NTD=%d" % ntd, "-DZEST_V_V2=%s" % v2] +
             # kernels with features: no SLP vectorizer.  It fuses the modulation multiplies of the row-block
             # epilogues into v_pk_mul_f32, which beside MFMAs costs more issue time than the two plain multiplies it
             # replaces (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'): +1.0 % / +1.8 % on the one- / two-net
             # feature kernels, -0.7 % on the featureless one, which keeps it (profiles/r03_ab_slp_ring.txt)
             (["-fno-slp-vectorize"] if (nts or ntd) else []) +
             # ... and with the AMDGPU-specific register-pressure trackers in the scheduler: +3.1 % (one net) / +4 % (two
             # nets) in two A/B pairs each in bf16, nothing on the featureless kernel (profiles/r03_ab_sched.txt); the split-fp16
             # kernels (at 256 VGPRs) answer with a few more spills and keep the default
             (["-mllvm", "-amdgpu-use-amdgpu-trackers=1"] if ((nts or ntd) and pt != "x3") else []) +
             # the featureless single-net kernel (the headline): LLVM's max-ILP scheduling strategy, +1.3 % in three A/B
             # pairs in bf16 (226 instead of 246 VGPRs, no scratch), +1.7 % in split fp16 (profiles/r03_ab_sched.txt); the
             # feature kernels lose 1 % (bf16) or gain nothing (split fp16) with it, the post-RA scheduler switched off
             # loses 1 - 3 %
             (["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if tag == "s0" else []))
            # (tried and not kept, profiles/r03_ab_mcache.txt: "-DZEST_CHUNK=8 -DZEST_SLOTS=8 -DZEST_AHEAD=4
            # -DZEST_MCACHE_JB=5" for the 16-bit kernels with two feature k-tiles - a 64 KiB weight ring and, in the LDS
            # that frees, 5 of the 8 row blocks of the modulation m per wave (mlp_engine.cuh): 9.7 % fewer MFMAs, +2.7 % /
            # +1.5 % on the one- / two-net kernels, but m rounded to 16 bits moves 0.7 % of the rays of the full-size
            # comparison past its bound, so the recompute in fp32 stays)
            for tag, nts, dyn, ntd, v2 in _SHAPES for pt, ep in _PRECS]
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


def _compile(src, extra, objname=None):
    obj = os.path.join(OBJ, (objname or src.replace(".hip", "")) + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _deps()):
        cmd = [HIPCC] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
        # register / scratch / LDS use of every kernel: kept next to the object (tools/kernel_resources.py)
        remarks = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
        with open(obj[:-2] + ".remarks", "w") as f:
            f.write("\n".join(remarks) + "\n")
        rest = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l
                and "loop not unrolled" not in l]
        if any(("warning" in l or "error" in l) for l in rest):
            sys.stderr.write("\n".join(rest) + "\n")
    return obj


def build(force=False, extra_flags=(), tag=None, only=None):
    """tag: build an experiment variant libzest_hip_<tag>.so with extra -D flags (objects kept
    apart); only: restrict to these sources (others are taken from the default build)."""
    global OBJ, LIB
    if tag:
        OBJ = os.path.join(CSRC, "build_" + tag)
        LIB = os.path.join(HERE, "libzest_hip_%s.so" % tag)
    os.makedirs(OBJ, exist_ok=True)
    every = [(name, "fused_variant.hip", flags) for name, flags in VARIANTS]
    every += [(s.replace(".hip", ""), s, SOURCE_FLAGS.get(s, [])) for s in SOURCES]
    jobs = [j for j in every if not only or j[0] in only or j[1] in only]
    if force:
        for name, _, _ in jobs:
            o = os.path.join(OBJ, name + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("ZEST_BUILD_JOBS", "8"))) as ex:
        objs = list(ex.map(lambda j: _compile(j[1], j[2] + list(extra_flags), j[0]), jobs))
    default_obj = os.path.join(CSRC, "build")
    objs += [os.path.join(default_obj, j[0] + ".o") for j in every if j not in jobs]
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lrocblas"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    # python build_hip.py [-f] [--tag NAME -DFOO=1 --flag=-fno-slp-vectorize ...] [--only obj1,obj2.hip,...]
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else None
    only = argv[argv.index("--only") + 1].split(",") if "--only" in argv else None
    print(build(force="-f" in argv, extra_flags=[a for a in argv if a.startswith("-D")] + [a[7:] for a in argv if a.startswith("--flag=")], tag=tag, only=only))
