#!/usr/bin/env python
"""Analyse kernel concurrency in a rocprofv3 --kernel-trace database (two-stream overlap)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, stream_id, queue_id from kernels where name like '%specmi%' order by start"))
print('kernels', len(rows), 'streams', sorted(set(r[3] for r in rows)), 'queues', sorted(set(r[4] for r in rows)))
# take the last 40% of the run (steady state)
t0 = rows[int(len(rows) * 0.6)][1]
rows = [r for r in rows if r[1] >= t0]
ev = []
for n, s, e, st, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
act = 0; last = ev[0][0]; tot = {}
for t, d in ev:
    tot[act] = tot.get(act, 0) + (t - last); last = t; act += d
span = ev[-1][0] - ev[0][0]
print('span ms', span / 1e6, {k: round(v / span, 3) for k, v in sorted(tot.items())})
bys = {}
for n, s, e, st, q in rows:
    bys.setdefault(st, []).append((s, e))
for st, iv in bys.items():
    print('stream', st, 'kernels', len(iv), 'busy ms', sum(e - s for s, e in iv) / 1e6)
