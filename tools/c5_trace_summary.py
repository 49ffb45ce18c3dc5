"""Condenses a rocprofv3 --kernel-trace CSV of tools/c5_probe.py: per kernel name count / total / mean duration, the span of the trace, the time during which
at least one kernel ran, and the average number of kernels in flight.   python tools/c5_trace_summary.py <kernel_trace.csv>"""
import csv, re, sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev, per = [], defaultdict(lambda: [0, 0.0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"k_\w+", r["Kernel_Name"])
        name = m.group(0) if m else r["Kernel_Name"][:40]
        per[name][0] += 1; per[name][1] += (e - s) / 1e3
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    t0, t1 = ev[0][0], ev[-1][0]
    busy = inflight_ns = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth > 0: busy += t - last
        inflight_ns += depth * (t - last)
        depth += d; last = t
    span = (t1 - t0) / 1e6
    print("span %.2f ms, >= 1 kernel running %.2f ms (%.0f %%), kernels in flight on average %.2f" % (span, busy / 1e6, 100 * busy / (t1 - t0), inflight_ns / (t1 - t0)))
    for name, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%-42s n=%6d total %9.1f us  mean %8.1f us" % (name, n, us, us / n))


if __name__ == "__main__":
    main()
