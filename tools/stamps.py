#!/usr/bin/env python3
"""Read the in-kernel stamps of a ZEST_STAMPS diagnostic build (build_hip.py --tag stamps
-DZEST_STAMPS): per-wave shader cycles spent in encode / engine / composite and, inside the
engine, in the per-chunk DMA wait + barrier and DMA issue.  Shares, not run times, are the
result (stamps forbid overlaps the product build has; cdna_hip_programming.md section 7).

    ZEST_HIP_LIB=$PWD/zest-nerf_amd/libzest_hip_stamps.so python tools/stamps.py [workload]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import zest_hip
import zest_renderer as renderer

wl = sys.argv[1] if len(sys.argv) > 1 else "nsff_static_1024x128"
d = bench.build_workload(wl, 1, torch.device("cuda:0"))
need = int(zest_hip.lib().zest_render_fused_workspace(d.R, d.S))
ws = torch.zeros(need, dtype=torch.uint8, device="cuda:0")
orig = zest_hip.render_fused
zest_hip.render_fused = lambda *a, **k: orig(*a, **dict(k, workspace=ws))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000       # ~0.3 s of sustained load before reading
with torch.no_grad():
    for _ in range(steps):
        bench.render_step(d)
    torch.cuda.synchronize()
stamp_bytes = 1024 * 8 * 8 * 8                 # csrc/fused.hip kStampBytes: the tail of the workspace
st = ws[need - stamp_bytes:].view(torch.int64).cpu().numpy().reshape(-1, 8)
st = st[st[:, 7] > 0]
tot, enc, eng, comp, wait, issue, real, n = [st[:, i].astype(np.float64) for i in range(8)]
vm, n = (st[:, 7] >> 16).astype(np.float64), (st[:, 7] & 0xFFFF).astype(np.float64)
clk = tot.mean() / (real.mean() / 100e6) / 1e9
print("%s: %d waves, passes/wave %.1f, wave lifetime %.0f cycles = %.1f us at %.2f GHz"
      % (wl, len(st), n.mean(), tot.mean(), real.mean() / 100, clk))
for name, v in (("encode", enc), ("engine", eng), ("  of which chunk wait+sync", wait),
                ("    of which own-DMA (vmcnt) wait", vm),
                ("  of which DMA issue", issue), ("composite+store", comp)):
    print("  %-32s %9.0f cycles  %5.1f %%" % (name, v.mean(), 100 * v.mean() / tot.mean()))
