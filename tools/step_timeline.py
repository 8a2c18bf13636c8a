#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace of bench.py (lanes on, graphs on): where the chip is idle
or nearly idle.  For every kernel the trace has start / end and the grid; the script picks a steady-state step (the span
between two consecutive k_adam launches), merges the intervals and prints
  * the step span, the union of kernel-busy time, the gaps (no kernel at all) longer than 2 us with their neighbours,
  * stretches where only "small" kernels run (fewer than 128 workgroups in flight), by the kernels involved.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $REPO/bench.py --steps 6 \
                  --warmup 2 --no-cpu-baseline --engine-only ; python $REPO/tools/step_timeline.py /tmp/tr
"""
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    rows = []
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            wg = 1
            for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
                wg *= max(int(r.get(k, 1) or 1), 1)
            ws = 1
            for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"):
                ws *= max(int(r.get(k, 1) or 1), 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], wg // max(ws, 1)))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if len(adam) < 4:
        print("not enough steps in the trace")
        return
    a, b = adam[-3], adam[-2]                        # a steady-state step: after adam[-3] ... through adam[-2]
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
    # union of busy time and gaps
    busy, gaps, cur_end = 0, [], t0
    prev = rows[a][2]
    for s, e, name, blocks in step:
        if s > cur_end:
            gaps.append((s - cur_end, prev, name, (cur_end - t0)))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        if e >= cur_end:
            prev = name
    print(f"busy (union) {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us in {len(gaps)} gaps")
    for g, p, n, at in sorted(gaps, reverse=True)[:12]:
        print(f"   gap {g / 1e3:6.1f} us at +{at / 1e3:7.1f} us   after {p:40s} before {n}")
    # low-occupancy stretches: sweep line over start / end events, sum of blocks of the kernels in flight
    ev = []
    for s, e, name, blocks in step:
        ev.append((s, 1, name, blocks)); ev.append((e, -1, name, blocks))
    ev.sort()
    live, last, low = {}, t0, {}
    for t, kind, name, blocks in ev:
        if live and sum(v[1] for v in live.values()) < 128 and t > last:
            key = " + ".join(sorted({v[0] for v in live.values()}))
            low[key] = low.get(key, 0) + (t - last)
        last = t
        if kind == 1:
            live[(name, t, blocks)] = (name, blocks)
        else:
            for k in list(live):
                if k[0] == name and k[2] == blocks:
                    del live[k]
                    break
    tot = sum(low.values())
    print(f"time with < 128 workgroups launched among the kernels in flight: {tot / 1e3:.1f} us")
    for k, v in sorted(low.items(), key=lambda kv: -kv[1])[:25]:
        print(f"   {v / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
