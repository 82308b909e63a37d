#!/bin/bash
# PMC passes for one kernel of a python command (run through gpurun from the repo root):
#   bash tools/prof_kernel_pmc.sh <kernel-name-substring> <python script and args ...>
# Counter sets are kept small (they must fit one pass each); never combined with a trace domain.
R=${GRAFT_REPO_ROOT:-$PWD}; k=$1; shift
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/pk; rocprofv3 --output-format csv --pmc $set -d /tmp/pk -o pk -- python3 $R/"$@" > /tmp/pk.log 2>&1
  python3 - "$k" <<'PY'
import csv,glob,collections,sys
f=glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k,v in d.items()})
PY
done
