"""Idle time between the kernels of the per-timestep paths, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 experiments/per_step_timing.py 512
    python experiments/trace_gaps.py DIR
For every window of the trace whose kernels belong to net.test / net.learn timesteps: busy time (sum of kernel durations), span
(first start .. last end) and the gaps between consecutive kernels, grouped by the kernel that FOLLOWS the gap."""


def main():
    import csv, glob, os, sys
    from collections import defaultdict
    fn = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    # steady part: the last 60 % of the launches of each phase (test first, then learn: the learn phase has k_bwd_* kernels)
    first_bwd = next(i for i, r in enumerate(rows) if r[2].startswith("k_bwd"))
    for name, seg in (("test", rows[int(first_bwd * 0.4):first_bwd - 20]), ("learn", rows[first_bwd + int((len(rows) - first_bwd) * 0.4):])):
        busy = sum(e - s for s, e, _ in seg)
        span = seg[-1][1] - seg[0][0]
        gaps = defaultdict(list)
        for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
            gaps[n1[:40]].append(s1 - e0)
        nstep = sum(1 for r in seg if r[2].startswith("k_lif_step_c1"))
        print("%s: %d timesteps, span %.1f us / step, busy %.1f us / step, idle %.1f us / step (%.1f %%)" % (
            name, nstep, span / 1e3 / nstep, busy / 1e3 / nstep, (span - busy) / 1e3 / nstep, 100.0 * (span - busy) / span))
        for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
            print("    gap in front of %-42s n %4d  mean %6.2f us  total / step %6.2f us" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3 / nstep))


if __name__ == "__main__":
    main()
