"""idle time between consecutive kernels of a rocprofv3 kernel trace: python scripts/trace_gaps.py <kernel_trace.csv> [min_gap_us]
Prints the total busy / idle time over the timed steps of bench.py and the largest classes of gaps by (kernel before, kernel after)."""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("isaac::", "").replace("void ", "").split("(")[0][-48:]))
rows.sort()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
# the timed steps of bench.py: from the first k_find_matches after the warm-up's k_cigar_pack to the end of the last k_cigar_pack
packs = [r for r in rows if "k_cigar_pack" in r[2]]
lo = min(r[0] for r in rows if "k_find_matches" in r[2] and r[0] > packs[0][1])
t1 = packs[-1][1]
rows = [r for r in rows if r[0] <= t1]
if len(sys.argv) > 3:      # the timed window as a small file: start (ns from the window start), end, kernel
    with open(sys.argv[3], "w") as f:
        for s_, e_, n_ in rows:
            if s_ >= lo:
                f.write("%d,%d,%s\n" % (s_ - lo, e_ - lo, n_))
busy, idle, gaps = 0, 0, collections.Counter()
counts = collections.Counter()
end = None
for s, e, n in rows:
    if s < lo:
        end = max(end or 0, e); prev = n
        continue
    if end is not None and s > end:
        g = s - end
        idle += g
        if g >= min_gap:
            gaps[(prev, n)] += g; counts[(prev, n)] += 1
    busy += max(0, e - max(s, end or s))
    if end is None or e > end:
        end = e; prev = n
print("window %.1f ms: busy %.1f ms, idle %.1f ms" % ((t1 - lo) / 1e6, busy / 1e6, idle / 1e6))
# the kernels of the window (the timed steps only: no warm-up, no set-up): launches and average duration, to set against bench.py's HIP events
per = collections.defaultdict(list)
for s, e, n in rows:
    if s >= lo:
        per[n].append(e - s)
print("kernels of the window by total time (launches, average ms, total ms; the totals add up to more than the window when streams overlap):")
for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("%6d %9.3f %9.2f   %s" % (len(d), sum(d) / len(d) / 1e6, sum(d) / 1e6, n))
for (a, b), g in gaps.most_common(25):
    print("%8.2f ms in %4d gaps   %s -> %s" % (g / 1e6, counts[(a, b)], a, b))
