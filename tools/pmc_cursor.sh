cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_cursor; rm -rf $O; mkdir -p $O
for tag in off on; do
  extra=""; [ $tag = on ] && extra="--wgrad-cursor 4,2"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$tag -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline $extra > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
for tag in ("off", "on"):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for f in glob.glob("$O/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE": continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
            acc[k][2] = max(acc[k][2], float(r["Counter_Value"]))
    k = "k_gemm_tn_bf16_grp"
    print(tag, k, "launches", acc[k][1], "mean fetch MB raw", acc[k][0] / max(acc[k][1], 1) / 1024, "largest launch MB raw", acc[k][2] / 1024)
PY
