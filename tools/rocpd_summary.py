#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (CSV on stdout)."""
import sqlite3
import sys


def main(path, top=60):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('kernel,calls,total_us,avg_us,min_us,max_us,percent')
    for name, calls, tot, avg, mn, mx in rows[:top]:
        print('"{}",{},{:.1f},{:.2f},{:.2f},{:.2f},{:.2f}'.format(name[:160].replace('"', "'"), calls, tot / 1e3, avg / 1e3,
                                                                 mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print('"TOTAL ({} kernels, {} launches)",{},{:.1f},,,,100.0'.format(len(rows), sum(r[1] for r in rows),
                                                                     sum(r[1] for r in rows), total / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
