#!/usr/bin/env python3
"""GPU busy / idle per training step from a rocprofv3 --kernel-trace CSV (steps are delimited by novograd_apply_kernel, or by the
kernel-name prefix in TIMELINE_DELIM - the backbone workloads step with torch's Adam: use wavlm_conv0_stats_kernel, the first
kernel of their forward): per hardware queue busy time, union-busy time (any queue running a kernel) and idle time.  Usage:
    python tools/timeline_busy.py gpurun_out/prof6/<host>/<pid>_kernel_trace.csv [step_timeline.txt]
"""
import collections
import csv
import os
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    delim = os.environ.get("TIMELINE_DELIM", "novograd_apply")
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(delim)]
    spans, unions, sums, mains = [], [], [], []
    for s in range(len(ends) // 2, len(ends) - 2):          # steady-state steps of the timed region
        seg = rows[ends[s] + 1:ends[s + 1] + 1]
        t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
        byq = collections.defaultdict(int)
        for r in seg:
            byq[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
        u, (cs, ce) = 0, iv[0]
        for a, b in iv[1:]:
            if a > ce:
                u += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        u += ce - cs
        spans.append((t1 - t0) / 1e6); unions.append(u / 1e6); sums.append(sum(byq.values()) / 1e6); mains.append(max(byq.values()) / 1e6)
    n = len(spans)
    med = lambda v: sorted(v)[len(v) // 2]
    if len(sys.argv) > 2:                                  # one steady-state step, kernel by kernel: start offset, duration, queue
        st = len(ends) * 3 // 4
        seg = rows[ends[st] + 1:ends[st + 1] + 1]
        t0 = int(seg[0]["Start_Timestamp"])
        qs = sorted({r["Queue_Id"] for r in seg})
        with open(sys.argv[2], "w") as f:
            last_end = {q: t0 for q in qs}
            for r in seg:
                a, b, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
                f.write(f"{(a - t0) / 1e3:9.1f} us  {(b - a) / 1e3:7.1f} us  q{qs.index(q)}  gap {(a - last_end[q]) / 1e3:6.1f}  "
                        f"{r['Kernel_Name'].split('(')[0][:70]}\n")
                last_end[q] = b
    print(f"{n} steps: span {med(spans):.3f} ms, union-busy {med(unions):.3f} ms ({100 * med(unions) / med(spans):.1f} %), "
          f"sum of kernel times {med(sums):.3f} ms, busiest queue {med(mains):.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
