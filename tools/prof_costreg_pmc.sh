#!/bin/bash
# PMC passes (HBM traffic, instruction mix) for the first layer of the regularisation net (tools/bench_costreg.py); separate passes, no trace domains
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/costreg_prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pk; rocprofv3 --output-format csv --pmc $set -d /tmp/pk -o pk -- python3 $R/tools/bench_costreg.py --passes ${1:-1} > /tmp/pk.log 2>&1
  python3 - <<'PY'
import csv,glob,collections
f=glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    for tag in ("conv3d_mfma_kernel<48", "deconv3d_mfma_kernel<16", "conv3d_mfma_kernel<16, 16"):
        if tag in k: d[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag,v in d.items(): print(tag, {k: round(sum(x)/len(x)) for k,x in v.items()})
PY
done | tee $out/r03_costreg_pmc_passes${1:-1}.txt
