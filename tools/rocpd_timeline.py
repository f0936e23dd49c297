#!/usr/bin/env python
"""Timeline view of a rocprofv3 rocpd kernel trace of bench.py: per training step (delimited by the optimizer kernel)
the wall time, the time at least one kernel was running, the summed kernel time (> wall when streams overlap) and
the phase split forward | losses | backward+optimizer (delimited by the CE kernels)."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.execute('select * from kernels limit 1')
    cols = [c[0] for c in cur.description]
    scol = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
    ecol = 'end' if 'end' in cols else [c for c in cols if c.startswith('end') or c.endswith('end')][0]
    qcol = next((c for c in ('stream_id', 'queue_id', 'stream', 'queue') if c in cols), None)
    rows = con.execute('select name, {}, {}{} from kernels order by {}'.format(
        scol, ecol, ', ' + qcol if qcol else '', scol)).fetchall()
    print('# columns:', cols)
    opt = [i for i, r in enumerate(rows) if 'optim_ema_kernel' in r[0]]
    print('# {} kernels, {} optimizer launches'.format(len(rows), len(opt)))
    print('step,wall_ms,busy_ms,sum_kernel_ms,fwd_ms,loss_ms,bwd_opt_ms,launches,queues')
    for si in range(1, len(opt)):
        seg = rows[opt[si - 1] + 1:opt[si] + 1]
        t0, t1 = rows[opt[si - 1]][2], seg[-1][2]
        # union of busy intervals
        busy, cur_s, cur_e = 0, None, None
        for r in sorted(seg, key=lambda r: r[1]):
            s, e = max(r[1], t0), r[2]
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += (cur_e - cur_s) if cur_e is not None else 0
        ce_f = [r for r in seg if 'ce_fwd_kernel' in r[0]]
        ce_b = [r for r in seg if 'ce_bwd' in r[0] or 'cons_bwd' in r[0]]
        fwd_end = ce_f[0][1] if ce_f else t0
        loss_end = max(r[2] for r in ce_b) if ce_b else fwd_end
        qs = len(set(r[3] for r in seg)) if qcol else 0
        print('{},{:.3f},{:.3f},{:.3f},{:.3f},{:.3f},{:.3f},{},{}'.format(
            si, (t1 - t0) / 1e6, busy / 1e6, sum(r[2] - r[1] for r in seg) / 1e6, (fwd_end - t0) / 1e6,
            (loss_end - fwd_end) / 1e6, (t1 - loss_end) / 1e6, len(seg), qs))


if __name__ == '__main__' and not (len(sys.argv) > 2 and sys.argv[2] == 'gaps'):
    main(sys.argv[1])


def gaps(path, step_index=-2, min_gap_us=15.0, top=40):
    """Idle gaps (no kernel running on any queue) inside one step + per-queue busy time per phase."""
    con = sqlite3.connect(path)
    rows = con.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
    opt = [i for i, r in enumerate(rows) if 'optim_ema_kernel' in r[0]]
    si = step_index if step_index >= 0 else len(opt) + step_index
    seg = rows[opt[si - 1] + 1:opt[si] + 1]
    t0 = rows[opt[si - 1]][2]
    ce = [r for r in seg if 'ce_fwd_kernel' in r[0]]
    fwd_end = ce[0][1] if ce else seg[-1][2]
    print('# step', si, 'kernels', len(seg), 'wall_ms', (seg[-1][2] - t0) / 1e6, 'fwd_ms', (fwd_end - t0) / 1e6)
    for phase, lo, hi in (('fwd', t0, fwd_end), ('bwd', fwd_end, seg[-1][2])):
        per_q = {}
        for r in seg:
            if r[1] >= lo and r[1] < hi:
                per_q.setdefault((r[3], r[4]), [0.0, 0])
                per_q[(r[3], r[4])][0] += (r[2] - r[1]) / 1e6
                per_q[(r[3], r[4])][1] += 1
        print('# phase', phase, 'wall_ms %.3f' % ((hi - lo) / 1e6), {k: (round(v[0], 3), v[1]) for k, v in per_q.items()})
    cur_e, last = t0, 'optim_ema_kernel(prev step)'
    out = []
    for r in seg:
        if r[1] > cur_e + min_gap_us * 1e3:
            out.append(((r[1] - cur_e) / 1e3, (cur_e - t0) / 1e6, last[:60], r[0][:60]))
        if r[2] > cur_e:
            cur_e, last = r[2], r[0]
    out.sort(reverse=True)
    print('# idle gaps > %g us: %d, total %.3f ms' % (min_gap_us, len(out), sum(g[0] for g in out) / 1e3))
    for g in out[:top]:
        print('gap_us %.1f at_ms %.3f after [%s] before [%s]' % g)


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'gaps':
    gaps(sys.argv[1])
