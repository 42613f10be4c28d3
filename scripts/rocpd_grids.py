"""Workgroups dispatched per kernel of ONE step from a rocprofv3 rocpd database (kernel trace): how many workgroups each launch puts
through the dispatcher, next to its duration -- launches whose workgroups mostly return in their prologue (launch gates, ladder levels)
show as many workgroups and a few microseconds.
    rocpd_grids.py results.db <name of the kernel that starts a step>"""
import sqlite3, sys


def main(path, first):
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
    g = [c for c in cols if "grid" in c.lower()]
    w = [c for c in cols if "workgroup" in c.lower()]
    print("columns:", cols)
    sel = ", ".join(g + w)
    rows = db.execute("select name, start, end, %s from kernels order by start" % sel).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    i0 = starts[-2] if len(starts) > 1 else starts[-1]
    i1 = starts[-1] if len(starts) > 1 else len(rows)
    agg = {}
    for r in rows[i0:i1]:
        name, s, e = r[0], r[1], r[2]
        dims = r[3:]
        gd, wd = dims[:len(g)], dims[len(g):]
        threads = 1
        for v in gd:
            threads *= max(int(v), 1)
        wsz = 1
        for v in wd:
            wsz *= max(int(v), 1)
        wgs = threads // max(wsz, 1)
        a = agg.setdefault(name.split("(")[0][-70:], [0, 0, 0.0])
        a[0] += 1; a[1] += wgs; a[2] += (e - s) / 1e3
    tot = 0
    for k, (n, wgs, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s launches %2d  workgroups %8d  %8.1f us" % (k, n, wgs, us))
        tot += wgs
    print("workgroups per step:", tot)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
