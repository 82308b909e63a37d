#!/bin/bash
# A/B of the encoding-volume layout on the GPU box: the round-2 library (plane-major [D][H][W][8]; build it from
# commit ecdb515 in a git worktree and copy its libzest_hip.so to
# zest-nerf_amd/libzest_hip_r02layout.so during round 3) against the current one ([H][W][D][8], depth innermost),
# same process order, feature workloads with random and with coherent (pixel-grid) rays.
for wl in nsff_static_mvs_1024x128 nsff_static_mvs_grid_1024x128 nsff_zest_val_1024x128 nsff_zest_val_grid_1024x128; do
  for t in r02layout base r02layout base; do
    lib=zest-nerf_amd/libzest_hip_$t.so; [ "$t" = base ] && lib=zest-nerf_amd/libzest_hip.so
    r=$(ZEST_HIP_LIB=$PWD/$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-modes --workload $wl 2>/dev/null | tail -1)
    echo "$wl $t $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("rays/s=%.3fM ms=%.4f kernel_ms=%.4f frac=%.3f"%(d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))' 2>/dev/null || echo FAILED)"
  done
done
