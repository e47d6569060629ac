"""Overlap analysis of a rocprofv3 --kernel-trace CSV: over the last `ms` milliseconds of the trace (whole steps of a replayed /
eager loop), how much of the wall time has 0 / 1 / 2 / 3+ kernels in flight, and per kernel name the EXCLUSIVE time (it is the
only kernel running: an upper bound of what shortening it returns) beside its summed duration.
usage: python tools/trace_overlap.py kernel_trace.csv window_ms [steps_in_window]"""
import collections, csv, sys
path, win_ms = sys.argv[1], float(sys.argv[2])
nsteps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
t_end = max(r[1] for r in rows)
t0 = t_end - int(win_ms * 1e6)
rows = [r for r in rows if r[0] >= t0]
ev = []
for i, (s, e, n, q) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active, last = set(), t0
conc = collections.Counter(); excl = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
pair2 = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        k = len(active)
        conc[min(k, 4)] += dt
        if k == 1:
            excl[rows[next(iter(active))][2]] += dt
        last = t
    if d == 1: active.add(i)
    else: active.discard(i)
for s, e, n, q in rows:
    tot[n] += e - s; cnt[n] += 1
wall = (t_end - t0)
print("window %.2f ms (%g steps): kernels %d, summed %.2f ms/step, queues %d" % (wall / 1e6, nsteps, len(rows) / nsteps, sum(tot.values()) / 1e6 / nsteps, len(set(r[3] for r in rows))))
for k in sorted(conc):
    print("  %s kernels in flight: %7.2f ms/step (%4.1f %%)" % (("%d" % k) if k < 4 else "4+", conc[k] / 1e6 / nsteps, 100.0 * conc[k] / wall))
print("exclusive time by kernel (ms/step) | summed (ms/step) | launches/step")
for n, v in sorted(excl.items(), key=lambda kv: -kv[1])[:45]:
    print("  %8.3f | %8.3f | %6.1f  %s" % (v / 1e6 / nsteps, tot[n] / 1e6 / nsteps, cnt[n] / nsteps, n[:110]))
