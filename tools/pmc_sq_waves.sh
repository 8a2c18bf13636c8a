#!/bin/bash
# Where the waves' cycles go, per kernel (round 2's table, re-measured at HEAD): one rocprofv3 --pmc pass of the bench.
#   usage (on the GPU box): tools/pmc_sq_waves.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/pmc_sq_waves}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES \
  --kernel-trace --output-format csv -d $O/raw -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box --lanes 0 > $O/run.log 2>&1
python - <<PY
import csv, glob, collections
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(O + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
with open(O + "/sq_waves.csv", "w", newline="") as fh:
    cw = csv.writer(fh)
    cw.writerow(["kernel", "launches", "wave_cycles_per_launch_M", "wait_any", "wait_inst_any", "wait_inst_lds", "active_inst_any", "lds_idx_active_per_busy_cu_cycle", "lds_conflict_share"])
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:16]:
        a = acc[k]; w = max(a["SQ_WAVE_CYCLES"], 1.0)
        cw.writerow([k, n[k], f"{w / max(n[k], 1) / 1e6:.1f}", f"{a['SQ_WAIT_ANY'] / w:.3f}", f"{a['SQ_WAIT_INST_ANY'] / w:.3f}", f"{a['SQ_WAIT_INST_LDS'] / w:.3f}",
                     f"{a['SQ_ACTIVE_INST_ANY'] / w:.3f}", f"{a['SQ_LDS_IDX_ACTIVE'] / max(a['SQ_BUSY_CU_CYCLES'], 1.0):.3f}", f"{a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1.0):.4f}"])
print(open(O + "/sq_waves.csv").read())
PY
