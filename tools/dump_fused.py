"""Render one bench workload with the fused kernel and save the packed per-ray maps (debug aid:
run under two ZEST_HIP_LIB builds and diff)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
wl, out = sys.argv[1], sys.argv[2]
d = bench.build_workload(wl, 1234, torch.device("cuda:0"))
import zest_hip
ws = torch.zeros(int(zest_hip.lib().zest_render_fused_workspace(d.R, d.S)), device="cuda:0", dtype=torch.uint8)
_orig = zest_hip.render_fused
def _patched(*a, **k):
    k["workspace"] = ws
    return _orig(*a, **k)
zest_hip.render_fused = _patched
with torch.no_grad():
    for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
        m = bench.render_step(d)["zest_packed_maps"]
    torch.cuda.synchronize()
np.save(out, m.cpu().numpy())
rec = ws.view(torch.float32)[: d.R * ((d.S + 31) // 32) * 20].view(-1, 20).cpu().numpy()
dbg = rec[rec[:, 16] > 0][:, 12:20]
print("debug records", len(dbg)); print(dbg[:24])
