import os, sys, json, subprocess
# A/B the pass shapes on the configs[3] batch (4096 x 192, full and the 512-ray shard) and the headline batch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "zest-nerf_amd"))
import torch, bench, zest_hip
dev = torch.device("cuda:0")
for wl, rays in (("zest_val_4096x192", None), ("zest_val_4096x192", 512), ("nsff_static_1024x128", None), ("nsff_zest_val_1024x128", None)):
    d = bench.build_workload(wl, 1, dev, rays=rays)
    for shape in ("dense", "ranges", None, "dense", "ranges"):
        zest_hip.set_fused_passes(shape)
        with torch.no_grad():
            for _ in range(20): bench.render_step(d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): bench.render_step(d)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(wl, d.R, shape, "ms=%.4f" % ms, "Mrays/s=%.3f" % (d.R / ms / 1e3), zest_hip.fused_pass_shape(d.R, d.S), flush=True)
zest_hip.set_fused_passes(None)
