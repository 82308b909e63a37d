"""Host/GPU cost of the per-step all-gather (1-rank NCCL rehearsal) and of a hipGraph replay of the step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import zest_parallel
d = bench.build_workload(os.environ.get("WL", "nsff_static_1024x128"), 1, dev)

def timeit(fn, n=300, tag=""):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-34s host %.1f us  total %.1f us" % (tag, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)

with torch.no_grad():
    timeit(lambda: bench.render_step(d), tag="render")
    maps = bench.render_step(d)["zest_packed_maps"]
    pend = []
    def g():
        pend.append(zest_parallel.gather_maps(maps, d.R, async_op=True, force=True))
        if len(pend) > 2: pend.pop(0)[1].wait()
    timeit(g, tag="async gather only")
    def rg():
        maps = bench.render_step(d)["zest_packed_maps"]
        pend.append(zest_parallel.gather_maps(maps, d.R, async_op=True, force=True))
        if len(pend) > 2: pend.pop(0)[1].wait()
    timeit(rg, tag="render + async gather")
    works = []
    def rg_nowait():
        maps = bench.render_step(d)["zest_packed_maps"]
        works.append(zest_parallel.gather_maps(maps, d.R, async_op=True, force=True))
        if len(works) > 64: del works[:32]
    timeit(rg_nowait, tag="render + async gather, no wait")
    side = torch.cuda.Stream()
    buf = torch.empty(d.R, 16, device=dev)
    def r_sidecopy():
        maps = bench.render_step(d)["zest_packed_maps"]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf[:, :maps.shape[1]].copy_(maps)
    timeit(r_sidecopy, tag="render + side-stream copy")
    def rs():
        maps = bench.render_step(d)["zest_packed_maps"]
        zest_parallel.gather_maps(maps, d.R, force=True)
    timeit(rs, tag="render + sync gather")
    # graph replay
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): bench.render_step(d)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = bench.render_step(d)["zest_packed_maps"]
    timeit(gr.replay, tag="graph replay")
    def gg():
        gr.replay()
        pend.append(zest_parallel.gather_maps(out, d.R, async_op=True, force=True))
        if len(pend) > 2: pend.pop(0)[1].wait()
    timeit(gg, tag="graph replay + async gather")
dist.destroy_process_group()
