#!/bin/bash
# PMC passes for the plane-sweep kernel (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/ps; rocprofv3 --output-format csv --pmc $set -d /tmp/ps -o ps -- python3 $R/tools/bench_ops.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/ps/**/*counter_collection.csv", recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "volume_cost" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k,v in d.items()})
PY
done
