#!/bin/bash
# Full measurement pass for profiles/: bench line (+per-op), rocprofv3 kernel stats (serial plan order,
# --lanes 0 - the library default since the end of round 3 - so that per-kernel durations are not inflated by
# side-lane overlap), PMC HBM traffic.
# Then the per-op roofline table and the sampler bench.
# usage (on the GPU box): tools/measure_round.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-rXX}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 30 --warmup 5 --per-op $O/per_op_ms.txt > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-box --lanes 0 > $O/stats.log 2>&1
# the same with every GEMM as its own launch (chained launches off): the per-kernel durations of rounds 1-4, for continuity
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-box --lanes 0 --nt-chain 0 > $O/stats_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box --lanes 0 > $O/pmc_sq.log 2>&1
python - <<PY
import csv, glob, collections, shutil
O = "$O"
st = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], O + "/kernel_stats.csv")
st = glob.glob(O + "/stats_serial/**/*kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], O + "/kernel_stats_serial.csv")
def pmc(d, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
with open(O + "/pmc_hbm_traffic.csv", "w", newline="") as fh:
    cw = csv.writer(fh)
    cw.writerow(["kernel", "launches", "fetch_MB_raw", "fetch_MB_corrected_x2", "write_MB"])
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        n = fe[k][1]
        # counters are in KB; FETCH_SIZE under-reports wide streaming reads 2x on gfx950 (MI355X_MICROARCH.md)
        f = fe[k][0] / n / 1024.0
        w = wr[k][0] / max(wr[k][1], 1) / 1024.0 if k in wr else 0.0
        cw.writerow([k, n, f"{f:.2f}", f"{2*f:.2f}", f"{w:.2f}"])
# MFMA utilisation per kernel: MFMA-busy cycles (summed over the 1024 SIMDs) / (1024 x kernel cycles)
sq = {c: pmc("pmc_sq", c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
with open(O + "/pmc_mfma_util.csv", "w", newline="") as fh:
    cw = csv.writer(fh)
    cw.writerow(["kernel", "launches", "mfma_busy_cycles_per_launch", "gui_active_per_launch", "mfma_util_of_1024_simds",
                 "lds_conflict_share"])
    for k in sorted(sq["SQ_VALU_MFMA_BUSY_CYCLES"], key=lambda k: -sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][0]):
        n = sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][1]
        mf = sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][0] / n
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
        ga = sq["GRBM_GUI_ACTIVE"][k][0] / max(sq["GRBM_GUI_ACTIVE"][k][1], 1) / 8.0
        lc = sq["SQ_LDS_BANK_CONFLICT"][k][0] / max(sq["SQ_LDS_IDX_ACTIVE"][k][0], 1.0)
        cw.writerow([k, n, f"{mf:.0f}", f"{ga:.0f}", f"{mf / (1024.0 * ga):.4f}" if ga > 0 else "", f"{lc:.4f}"])
print(open(O + "/bench.json").read()[:600])
PY
# ties the PMC table to the kernel sources it was measured on (bench.py only uses a table whose sha matches its build)
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_sha())" > $O/pmc_hbm_traffic.sha
# per-op MFMA / HBM view of the step and the sampler's throughput table
cd $R
python tools/op_roofline.py $O/per_op_ms.txt > $O/op_roofline.txt 2>&1
python tools/bench_sampler.py --steps 3000 --batches 1,4,16,32 > $O/sampler.txt 2>&1
tail -5 $O/sampler.txt
