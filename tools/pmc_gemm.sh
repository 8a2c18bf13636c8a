#!/bin/bash
# PMC passes over the G1-shaped GEMM (tools/ablate_gemm.py).  usage: pmc_gemm.sh <WAVE_ROWS> <PIPE> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; WR=$1; PP=$2; TAG=$3
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  WAVE_ROWS=$WR PIPE=$PP ABL=0 timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python $R/tools/ablate_gemm.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_gemm_nt" not in r["Kernel_Name"]: continue
            k = r["Counter_Name"]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in acc.items():
            print(f"$TAG {k:28s} per-launch {v / max(n,1):16.1f}  (n={n})")
PY
