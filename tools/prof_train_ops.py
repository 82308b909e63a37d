"""Which torch operators (not zest kernels) run in one bf16 training step of tools/bench_train.py's step, by autograd node / op name
and input shapes: the launches that item 4(b) of the round-2 review asks to remove."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import zest_renderer as renderer
dev = torch.device("cuda:0")
d = bench.build_workload("nsff_zest_val_1024x128", 5, dev, 1024)
d.args.precision, d.args.zest_maps_only = 16, False
vol_s, vol_d = d.vol_s.clone().requires_grad_(True), d.vol_d.clone().requires_grad_(True)
params = list(d.net_s.parameters()) + list(d.net_d.parameters())


def step(loss_too=True):
    for p in params:
        p.grad = None
    vol_s.grad = vol_d.grad = None
    ret = renderer.rendering(
        d.args, d.t["rays_pts"], d.t["rays_ndc"], d.t["depth_candidates"], d.t["rays_dir"],
        volume_feature_static=vol_s, volume_feature_dynamic=vol_d, imgs=d.imgs, neighbour_frames=d.nb_imgs,
        im_cam_mat=d.cam, nb_cam_mat=d.nb_cam, network_fn=d.net_s, network_fn_dy=d.net_d,
        embedding_pts=d.emb[0], embedding_xyzt=d.emb[1], embedding_dir=d.emb[2], ref_frame_idx=0.1,
        num_frames=24, scene_flow=True, val=False, chain_5frames=False, raw_noise_std=0)
    with torch.profiler.record_function("HARNESS_LOSS"):
        loss = sum(v.square().mean() for k, v in ret.items()
                   if v is not None and v.requires_grad and k not in ("raw_rgba", "input_feat"))
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("aten ops with device time: %d calls, %.1f us" % (sum(e.count for e in rows), tot))
for e in rows[:45]:
    print("%8.1f us x%-3d %-28s %s" % (e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
