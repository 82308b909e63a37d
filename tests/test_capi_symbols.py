"""CPU: the C-ABI library loads and exports every symbol include/zest_render.h declares."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zest_render.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zest_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    import zest_hip
    assert sorted(zest_hip.exported_symbols()) == _declared()


def test_library_exports_every_declared_symbol():
    import zest_hip
    if not os.path.exists(zest_hip.LIB_PATH):
        import build_hip
        build_hip.build()
    lib = ctypes.CDLL(zest_hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.zest_abi_version() == 3


def test_missing_library_is_an_error(monkeypatch):
    import zest_hip
    monkeypatch.setattr(zest_hip, "_lib", None)
    monkeypatch.setattr(zest_hip, "LIB_PATH", "/nonexistent/libzest_hip.so")
    with pytest.raises(RuntimeError, match="no non-HIP execution path"):
        zest_hip.lib()


def test_cpu_tensors_are_rejected():
    import torch
    import zest_hip
    with pytest.raises(RuntimeError, match="runs only on a HIP device"):
        zest_hip.embed(torch.zeros(4, 3), 10)


def test_library_is_not_older_than_its_sources():
    """A failed rebuild must not go unnoticed: the shipped .so is at least as new as every source it is
    built from (build_hip.py rebuilds what is stale; __graft_entry__.build() runs it)."""
    import glob
    pkg = os.path.join(ROOT, "zest-nerf_amd")
    lib = os.path.join(pkg, "libzest_hip.so")
    srcs = glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.cuh")) + \
        glob.glob(os.path.join(pkg, "csrc", "*.h")) + [os.path.join(ROOT, "include", "zest_render.h")]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(lib) >= os.path.getmtime(newest), "libzest_hip.so is older than %s: rebuild" % newest


def test_fused_pass_shapes_for_the_baseline_batches():
    """Host arithmetic of the fused renderer's pass shape (no GPU call: cus given): (ray ranges?, workgroups, rounds).
    Every baseline batch finishes its rays in the kernel (a range of whole rays per workgroup, no block records in
    HBM, no second launch) with a block for every wave of every pass but a workgroup's last: the strong-scaled
    configs[3] shard (512 rays x 192 samples per GPU: 6 blocks per ray) runs an 8-block and a 4-block pass per
    workgroup, the full 4096 x 192 batch 12 rounds (whole-ray passes: 16)."""
    sys.path.insert(0, os.path.join(ROOT, "zest-nerf_amd"))
    import zest_hip as zh
    assert zh.fused_pass_shape(1024, 128, zh.PREC_BF16, 256) == (1, 256, 2)        # 4 rays = 16 blocks per workgroup
    assert zh.fused_pass_shape(1024, 128, zh.PREC_F16X3, 256) == (1, 256, 4)       # 16-sample blocks: 8 per ray
    assert zh.fused_pass_shape(512, 192, zh.PREC_BF16, 256) == (1, 256, 2)         # 2 rays = 12 blocks: passes of 8 + 4
    assert zh.fused_pass_shape(4096, 192, zh.PREC_BF16, 256) == (1, 256, 12)       # 16 rays = 96 blocks: 12 full passes
    assert zh.fused_pass_shape(8192, 128, zh.PREC_F16, 256) == (1, 256, 16)
    assert zh.fused_pass_shape(100, 64, zh.PREC_BF16, 256) == (1, 100, 1)          # fewer rays than CUs: one each
    assert zh.fused_pass_shape(7, 300, zh.PREC_BF16, 256) == (0, 9, 1)             # a few long rays: spread the blocks
    assert zh.fused_pass_shape(600, 300, zh.PREC_BF16, 256) == (0, 250, 3)         # ranges would need a 4th round
    assert zh.fused_pass_shape(2048, 300, zh.PREC_BF16, 256) == (1, 256, 10)       # 10 blocks per ray, carried over passes
    assert zh.fused_pass_shape(0, 64, zh.PREC_BF16, 256)[1:] == (0, 0)


def test_fused_pass_shape_covers_every_block_once():
    """The host's choice (workgroups, rounds) against a restatement of the kernel's block assignment
    (csrc/fused.cuh: ray ranges [w R / n, (w+1) R / n) x blocks per ray; dense: equal shares of whole passes):
    every block belongs to exactly one workgroup, no workgroup is empty, `rounds` is the busiest workgroup's pass
    count, and ray ranges are never taken when they need more rounds than the dense shape."""
    sys.path.insert(0, os.path.join(ROOT, "zest-nerf_amd"))
    import random
    import zest_hip as zh
    rnd = random.Random(7)
    cases = [(1, 1, 256), (255, 33, 256), (256, 128, 256), (257, 128, 256), (4097, 192, 304), (5, 1000, 8)]
    cases += [(rnd.randint(1, 6000), rnd.randint(1, 700), rnd.choice([8, 64, 104, 256, 304])) for _ in range(200)]
    for prec, bs in ((zh.PREC_BF16, 32), (zh.PREC_F16X3, 16)):
        for R, S, cus in cases:
            ranges, n_wg, rounds = zh.fused_pass_shape(R, S, prec, cus)
            bpr = -(-S // bs)
            n_blocks = R * bpr
            assert 1 <= n_wg <= min(cus, n_blocks)
            if ranges:
                assert n_wg == min(cus, R)
                spans = [((w * R // n_wg) * bpr, ((w + 1) * R // n_wg) * bpr) for w in range(n_wg)]
            else:
                share = -(-(-(-n_blocks // 8)) // n_wg) * 8
                spans = [(w * share, min((w + 1) * share, n_blocks)) for w in range(n_wg)]
            assert spans[0][0] == 0 and spans[-1][1] == n_blocks
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(b > a for a, b in spans)
            assert rounds == max(-(-(b - a) // 8) for a, b in spans)
            dense_rounds = -(-(-(-n_blocks // 8)) // min(cus, -(-n_blocks // 8)))
            assert rounds <= dense_rounds or not ranges


def test_train16_size_queries_refuse_other_mlp_shapes():
    """The bf16 training kernels are unrolled for depth 8 / width 256 / skips [4]; the size queries of the C ABI
    (no GPU call) answer 0 with an error text for any other descriptor, also AFTER a default-shape query has
    been served (the table cache is keyed on the whole shape)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "zest-nerf_amd"))
    import zest_hip as zh
    lib = zh.lib()
    good = zh.MlpDesc(63, 40, 27, 1, 0, zh.HEAD_NONE)
    assert lib.zest_mlp_train16_stash_bytes(C.byref(good), 4096) > 0
    assert lib.zest_mlp_train16_work_bytes(C.byref(good), 4096) > 0
    assert lib.zest_mlp_train16_packed_bytes(C.byref(good)) > 0
    for bad in (zh.MlpDesc(63, 40, 27, 1, 0, zh.HEAD_NONE, 5, 128, 1 << 2),
                zh.MlpDesc(63, 40, 27, 1, 0, zh.HEAD_NONE, 8, 256, 1 << 3),
                zh.MlpDesc(63, 40, 27, 1, 2, zh.HEAD_NONE)):                      # 'v2': no bf16 training kernel
        assert lib.zest_mlp_train16_stash_bytes(C.byref(bad), 4096) == 0
        assert b"zest_mlp_train16" in lib.zest_last_error()
        assert lib.zest_mlp_train16_work_bytes(C.byref(bad), 4096) == 0
        assert lib.zest_mlp_train16_packed_bytes(C.byref(bad)) == 0
