#!/usr/bin/env python
"""One training step of a rocprofv3 kernel trace (rocpd database of bench.py) as a readable timeline: kernels in start
order, consecutive launches of one kernel on one queue merged, with the time the GPU ran NOTHING in front of each group --
where the step is serial (losses, stem, optimizer, copies) and where two streams overlap.
    python tools/step_timeline.py <results.db> [step index]"""
import sqlite3
import sys


def short(name):
    n = name.replace('void ', '').replace('cms::', '')
    return n[:n.index('(')] if '(' in n else n


def main(path, which=-3):
    con = sqlite3.connect(path)
    cur = con.execute('select * from kernels limit 1')
    cols = [c[0] for c in cur.description]
    scol = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
    ecol = 'end' if 'end' in cols else [c for c in cols if c.startswith('end') or c.endswith('end')][0]
    qcol = next((c for c in ('stream_id', 'queue_id', 'stream', 'queue') if c in cols), None)
    rows = con.execute('select name, {}, {}{} from kernels order by {}'.format(scol, ecol, ', ' + qcol if qcol else '', scol)).fetchall()
    opt = [i for i, r in enumerate(rows) if 'optim_ema_kernel' in r[0]]
    si = which if which >= 0 else len(opt) + which
    seg = rows[opt[si - 1] + 1:opt[si] + 1]
    t0 = rows[opt[si - 1]][2]
    print('# step {}: {} kernels, wall {:.3f} ms'.format(si, len(seg), (seg[-1][2] - t0) / 1e6))
    print('# start_us  dur_us  idle_before_us  queue  n  kernel')
    busy_end = t0
    groups = []
    for name, s, e, *q in seg:
        q = q[0] if q else 0
        idle = max(0, s - busy_end)
        if groups and groups[-1][4] == short(name) and groups[-1][3] == q and idle == 0:
            groups[-1][1] = e
            groups[-1][5] += 1
        else:
            groups.append([s, e, idle, q, short(name), 1])
        busy_end = max(busy_end, e)
    tot_idle = 0
    for s, e, idle, q, name, n in groups:
        tot_idle += idle
        print('{:9.1f} {:8.1f} {:8.1f}   q{:<3} {:3d}  {}'.format((s - t0) / 1e3, (e - s) / 1e3, idle / 1e3, q, n, name[:90]))
    print('# idle (no kernel running) inside the step: {:.3f} ms'.format(tot_idle / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -3)
