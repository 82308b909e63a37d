#!/bin/bash
# A/B the tagged experiment builds of libzest_hip on the box: bench the headline workloads with each.
# usage: tools/ab_variants.sh "tag1 tag2 ..." [workload ...]
tags=$1; shift
wls=${@:-nsff_static_1024x128}
for wl in $wls; do
  for t in base $tags; do
    lib=zest-nerf_amd/libzest_hip_$t.so; [ "$t" = base ] && lib=zest-nerf_amd/libzest_hip.so
    r=$(ZEST_HIP_LIB=$PWD/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $wl 2>/dev/null | tail -1)
    echo "$wl $t $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("rays/s=%.3gM ms=%.4f frac=%.3f"%(d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"]))' 2>/dev/null || echo FAILED)"
  done
done
