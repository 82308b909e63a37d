import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import bench, torch
d = bench.build_workload("nsff_static_mvs_1024x128", 1, torch.device("cuda:0"))
with torch.no_grad():
    for _ in range(20): bench.render_step(d)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(200): bench.render_step(d)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("host per call %.1f us; with sync %.1f us"%((t1-t0)/200*1e6,(t2-t0)/200*1e6))
    pr=cProfile.Profile(); pr.enable()
    for _ in range(200): bench.render_step(d)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
