"""How much of the step do two kernels run at the same time?  From a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp / Queue_Id
per dispatch): busy time of the union of all kernel intervals, the sum of the durations, time with >= 2 kernels in flight, per queue.
usage: overlap_from_trace.py DIR_WITH_kernel_trace_csv [skip_fraction]"""
import csv, glob, os, sys, collections
root = sys.argv[1]; skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']))
rows.sort()
t_lo, t_hi = rows[0][0], rows[-1][1]
cut = t_lo + int((t_hi - t_lo) * skip)            # the second part of the run: steady-state steps
rows = [r for r in rows if r[0] >= cut]
ev = []
for s, e, q, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = two = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: two += t - last
    depth += d; last = t
tot = sum(e - s for s, e, _, _ in rows)
span = rows[-1][1] - rows[0][0]
perq = collections.Counter()
for s, e, q, _ in rows: perq[q] += e - s
print('window %.1f ms: %d kernels, sum of durations %.1f ms, GPU busy (union) %.1f ms = %.1f %% of the window, two or more kernels in flight %.1f ms = %.1f %% of the busy time'
      % (span / 1e6, len(rows), tot / 1e6, busy / 1e6, 100.0 * busy / span, two / 1e6, 100.0 * two / busy))
for q, v in perq.most_common(4): print('   queue %s: %.1f ms of kernels' % (q, v / 1e6))
