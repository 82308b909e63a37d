"""The generators (callers of the path, reference networks.py:355-720): host orchestration.

CPU: the chunk sharding of DyMVSNeRF_G.forward_val over a world-size-2 gloo group, with the ray
sampler and the renderer replaced by bookkeeping stand-ins (pixel ids in, functions of pixel ids
out), so only the partition / all-gather / result assembly is under test.
GPU: a full forward_val and forward on a small synthetic batch with frozen random-weight volume
builders - chunk-size invariance, result structure, fp32 and fused bf16 plans."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    a = SimpleNamespace(batch_size=64, N_samples=24, pad=4, vis_cnn=False, save_test=".", patch_size=-1,
                        scale_anneal=-1, gan_type=None, white_bkgd=False, with_chain_loss=True,
                        use_motion_mask=True, num_extra_samples=16, raw_noise_std=0.0, chunk=256,
                        img_downscale=1.0, netchunk=4096, feat_dim=20, feat_dim_dy=20,
                        use_color_volume=False, net_type="v0", precision=32)
    a.__dict__.update(kw)
    return a


def _shard_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "zest-nerf_amd"), os.path.join(ROOT, "tests"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zest_networks as networks
    import zest_renderer as renderer
    import zest_utils as utils
    H, W, chunk = 6, 9, 8                                  # 54 pixels: 7 chunks, the last one short
    calls = []

    def fake_rays(imgs, depths, w2cs, c2ws, intr, nf, S, chunk=-1, idx=-1, **kw):
        ids = torch.arange(idx * chunk, min((idx + 1) * chunk, H * W), dtype=torch.float32)
        calls.append((idx, chunk))
        assert kw.get("zest_rays_only") is True
        return (ids[None, :, None, None].expand(1, -1, S, 3), ids[None, :, None].expand(1, -1, 3), None,
                ids[None, :, None, None].expand(1, -1, S, 3), ids[None, :, None].expand(1, -1, S)) + (None,) * 6

    def fake_render(args, rays_pts, rays_ndc, z, rays_dir, **kw):
        ids = rays_dir[0, :, 0]
        assert kw["val"] and args.zest_maps_only
        out = {}
        for j, k in enumerate(networks.DyMVSNeRF_G.VAL_KEYS):
            v = ids * (j + 1)
            out[k] = (torch.stack([v, v + 0.25, v + 0.5], -1) if "rgb" in k else v)[None]
        return out
    utils.build_rays_dy, renderer.rendering = fake_rays, fake_render
    gen = networks.DyMVSNeRF_G(_args(chunk=chunk), 1, None, None, None, None, None, None, None)
    x = dict(images=torch.zeros(1, 3, 3, H, W), near_fars=torch.zeros(1, 3, 2), w2cs=None, intrinsics=None,
             c2ws=None, proj_mats=torch.zeros(1, 3, 3, 4), depths=None, time=torch.tensor(3), total_frames=torch.tensor(12),
             flow_fwds=None, flow_bwds=None, mask_fwds=None, mask_bwds=None)
    res = gen.forward_val(x)
    cat = [torch.cat(r) for r in res[1:]]
    q.put((rank, sorted(calls), [c.numpy() for c in cat], gen.chain_bwd, gen.args.zest_maps_only))
    dist.destroy_process_group()


def test_forward_val_shards_chunks_and_gathers_once():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # contiguous chunk runs, merged into as few sampler / renderer calls as the sampler's (index, size) addressing
    # allows: rank 0 renders chunks 0-3 as one chunk of 32 rays, rank 1 chunks 4-5 as chunk 2 of 16 rays, then chunk 6
    assert got[0][1] == [(0, 32)] and got[1][1] == [(2, 16), (6, 8)]
    ids = np.arange(54, dtype=np.float32)
    for rank, _, cat, chain_bwd, maps_only in got:
        assert chain_bwd is True and maps_only is False                   # 7 toggles; the flag is restored
        for j, c in enumerate(cat):
            want = ids * (j + 1)
            if c.ndim == 2:
                want = np.stack([want, want + 0.25, want + 0.5], -1)
            assert c.shape == want.shape and np.array_equal(c, want), (rank, j)


# ------------------------------------------------------------------------------------- GPU
def _batch(seed, H=32, W=32, V=3, V_dy=3):     # MVSNet's regulariser takes 3 views (32 + 9 channels)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_cases as gc
    import zest_synth as zs
    g = zs.rng(seed)
    inp = gc.rays_inputs(seed, V=V, H=H, W=W)
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cost = gc.cost_inputs(seed + 1, V=V + 1, H=H // 4, W=W // 4)
    cost_dy = gc.cost_inputs(seed + 2, V=V_dy, H=H // 4, W=W // 4)
    nb_w2cs, nb_intr = zs.make_cameras(V_dy, H, W, focal=30.0)
    return dict(images=G(inp["imgs"]), proj_mats=G(cost["proj_mats"]), near_fars=G(inp["near_fars"]),
                w2cs=G(inp["w2cs"]), c2ws=G(inp["c2ws"]), intrinsics=G(inp["intrinsics"]), depths=G(inp["depths"]),
                time=torch.tensor(3), total_frames=torch.tensor(12), flow_fwds=G(inp["flow_fwd"]),
                flow_bwds=G(inp["flow_bwd"]), mask_fwds=G(inp["mask_fwd"]), mask_bwds=G(inp["mask_bwd"]),
                motion_coords=[G(inp["motion_coords"])], nb_imgs=G(g.uniform(0, 1, size=(1, V_dy, 3, H, W)).astype(np.float32)),
                nb_proj_mats=G(cost_dy["proj_mats"]), nb_w2cs=G(nb_w2cs), nb_intr=G(nb_intr))


def _generator(args, train_builders=False):
    import zest_networks as networks
    import golden_cases as gc
    torch.manual_seed(1)
    mk = lambda P, F, static: networks.MVSNeRF(D=8, W=256, input_ch_pts=P, output_ch=4, input_ch_views=gc.PE_DIR,
                                               input_ch_feat=F, skips=[4], net_type="v0", sceneflow=True,
                                               static=static, use_mvs=True).cuda()
    enc, enc_dy = (networks.MVSNet().cuda().requires_grad_(train_builders),
                   networks.MVSNet().cuda().requires_grad_(train_builders))
    return networks.DyMVSNeRF_G(args, 1, mk(gc.PE_XYZT, 20, False), mk(gc.PE_PTS, 20, True), enc, enc_dy,
                                networks.Embedding(3, 10), networks.Embedding(4, 10), networks.Embedding(3, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [32, 16])
def test_forward_val_is_chunk_invariant(hip, precision):
    x = _batch(91)
    outs = []
    # chunks of 256 and 384 rays one launch each, and chunks of 128 merged four / three at a time (zest_val_rays)
    for chunk, merge in ((256, 0), (384, 0), (128, 512), (128, 384)):
        gen = _generator(_args(chunk=chunk, precision=precision, zest_val_rays=merge))
        res = gen.forward_val(x)
        assert len(res[1]) == {(256, 0): 4, (384, 0): 3, (128, 512): 2, (128, 384): 3}[(chunk, merge)]
        assert tuple(res[0].shape) == (1, 4, 3, 32, 32)
        outs.append([torch.cat(r) for r in res[1:]])
        assert [tuple(o.shape) for o in outs[-1]] == [(1024, 3), (1024,), (1024, 3), (1024,), (1024, 3), (1024,), (1024,)]
        assert all(torch.isfinite(o).all() for o in outs[-1])
    tol = 1e-5 if precision == 32 else 1e-4
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.gpu
def test_training_forward_returns_the_reference_keys(hip):
    x = _batch(92)
    gen = _generator(_args())
    ret = gen(x, step=0)
    for k in ("rgb_map", "rgb_map_ref", "weights_ref_dy", "raw_sf_ref2prev", "target_s", "depth_gt", "t_vals",
              "rays_flow_fwd_gt", "rays_mask_bwd_gt", "chain_bwd", "chain_5frames"):
        assert k in ret, k
    assert ret["chain_bwd"] is True and ret["chain_5frames"] is False
    R = ret["target_s"].shape[1]
    assert R == 64 + 16 and tuple(ret["rgb_map_ref"].shape) == (1, R, 3)   # motion-mask extras in early steps
    loss = ret["rgb_map_ref"].square().mean() + ret["rgb_map"].square().mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in gen.nerf_static.parameters())
    assert all(p.grad is None for p in gen.encoding_net.parameters())      # frozen volume builder


@pytest.mark.gpu
def test_training_step_reaches_the_volume_builders(hip):
    """The reference optimises generator.parameters() INCLUDING both MVSNets (train.py:270): with trainable
    builders a training forward + backward leaves finite, non-zero gradients on FeatureNet and CostRegNet of
    the static and of the dynamic builder (rendering -> encode backward -> 3-D CNN -> HIP plane-sweep
    backward -> 2-D CNN)."""
    x = _batch(93)
    gen = _generator(_args(), train_builders=True)
    ret = gen(x, step=0)
    loss = ret["rgb_map_ref"].square().mean() + ret["rgb_map"].square().mean() + ret["rgb_map_ref_dy"].square().mean()
    loss.backward()
    for enc in (gen.encoding_net, gen.encoding_net_dy):
        grads = {k: p.grad for k, p in enc.named_parameters()}
        assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
        assert grads["feature.conv0.0.conv.weight"].abs().max() > 0 and grads["cost_reg_2.conv0.conv.weight"].abs().max() > 0


@pytest.mark.gpu
def test_forward_val_builds_the_two_volumes_on_two_streams(hip):
    """Whole-image evaluation builds the static and the dynamic encoding volume side by side on two HIP streams
    (independent nets and images); the image is the one the serial order gives, bit for bit, also when the call is
    repeated (the second stream is kept)."""
    x = _batch(91)
    outs = []
    for overlap in (True, False, True):
        gen = _generator(_args(chunk=256, precision=16, zest_overlap_builders=overlap))
        res = gen.forward_val(x)
        assert ("_zest_side_stream" in gen.__dict__) == overlap
        outs.append([torch.cat(r) for r in res[1:]])
        if overlap:
            again = [torch.cat(r) for r in gen.forward_val(x)[1:]]
            gen.chain_bwd = False
            assert all(torch.equal(a, b) for a, b in zip(outs[-1], again))
    for other in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], other))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [32, 16])
def test_builder_graph_replay_equals_direct_calls(hip, precision):
    """The all-HIP volume builder is recorded as a HIP graph at its first whole-image call and replayed afterwards
    (args.zest_graph_builders, default on): the volumes are those of direct calls bit for bit, for new inputs too, the
    norms' step counters advance once per call, and a weight update makes the next call record again."""
    x, x2 = _batch(91), _batch(92)
    gen, ref = (_generator(_args(chunk=256, precision=precision, zest_graph_builders=g)) for g in (True, False))
    ref.load_state_dict(gen.state_dict())
    with torch.no_grad():
        for k, batch in enumerate((x, x, x2, x)):
            a, b = gen._scene(batch, bn_batch_stats=True), ref._scene(batch, bn_batch_stats=True)
            assert torch.equal(a["vol_s"], b["vol_s"]) and torch.equal(a["vol_d"], b["vol_d"]), k
        assert "_zest_builder_graphs" in gen.__dict__ and "_zest_builder_graphs" not in ref.__dict__
        bn_a, bn_b = gen.encoding_net.cost_reg_2.conv3.bn, ref.encoding_net.cost_reg_2.conv3.bn
        assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 4
        assert torch.equal(bn_a.running_var, bn_b.running_var)
        for g in (gen, ref):
            g.encoding_net.cost_reg_2.conv2.conv.weight.mul_(1.25)
        graph = gen._zest_builder_graphs[id(gen.encoding_net)]["graph"]
        a, b = gen._scene(x2, bn_batch_stats=True), ref._scene(x2, bn_batch_stats=True)
        assert torch.equal(a["vol_s"], b["vol_s"])
        assert gen._zest_builder_graphs[id(gen.encoding_net)]["graph"] is not graph
