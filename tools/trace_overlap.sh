#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trace_ov; mkdir -p $R/gpurun_out/trace_ov
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_ov -- python $R/tools/phase_times.py > $R/gpurun_out/trace_ov/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/trace_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the first k_copy_table with a huge grid (decoder pack) and print the next 30 kernels relative to it
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_copy_table") and int(r["Grid_Size_X"]) > 5_000_000]
i0 = idx[len(idx)//2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0-2:i0+28]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  q={r.get("Queue_Id","?")} grid={r["Grid_Size_X"]:>9} {r["Kernel_Name"][:40]}')
PY
