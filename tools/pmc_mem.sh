#!/bin/bash
# Memory-path PMC passes over the bench (serial plan order): what the gated-layer GEMM and the wgrad GEMM wait for.
# usage (GPU box): tools/pmc_mem.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-mem}; O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
i=0
for C in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr" \
         "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum SQ_VMEM_TA_ADDR_FIFO_FULL TCC_CYCLE_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --engine-only --lanes 0 > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(s in k for s in ("k_gemm_nt_bf16<1", "k_gemm_nt_bf16<3", "k_gemm_nt_bf16<0, false, 4", "k_gemm_tn_bf16<0>", "k_fn<1")): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print("==", k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:40s} per launch {v / max(n, 1):16.1f}   (n={n})")
PY
