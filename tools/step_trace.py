#!/usr/bin/env python3
"""One steady-state training step of a rocprofv3 kernel trace as a list: start offset (us), duration (us), workgroups,
kernel name - everything that runs between two consecutive k_adam launches.  Kernels shorter than --min us are folded
into one line per run of consecutive short kernels.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $REPO/bench.py --steps 6 \
                  --warmup 2 --no-cpu-baseline --engine-only ; python $REPO/tools/step_trace.py /tmp/tr [--min 12]
"""
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    tmin = float(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 12.0
    rows = []
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            wg = 1
            for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")):
                wg *= max(int(r.get(k, 1) or 1), 1) // max(int(r.get(w, 1) or 1), 1) or 1
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], wg))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if len(adam) < 4:
        print("not enough steps in the trace")
        return
    a, b = adam[-3], adam[-2]
    t0 = rows[a][1]
    print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us, {b - a} kernels")
    small = []

    def flush():
        if small:
            s0, e1 = small[0][0], max(x[1] for x in small)
            names = {}
            for x in small:
                names[x[2]] = names.get(x[2], 0) + 1
            print(f"{(s0 - t0) / 1e3:9.1f} {(e1 - s0) / 1e3:8.1f}        {len(small)} short: " +
                  ", ".join(f"{n} x{c}" for n, c in sorted(names.items(), key=lambda kv: -kv[1])[:6]))
            small.clear()
    for s, e, name, wg in rows[a + 1:b + 1]:
        if (e - s) / 1e3 < tmin:
            small.append((s, e, name))
            continue
        flush()
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {wg:6d} {name}")
    flush()


if __name__ == "__main__":
    main()
