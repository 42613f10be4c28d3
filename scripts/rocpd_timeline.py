"""Kernel timeline of the LAST step in a rocprofv3 rocpd database: start / end (us, relative to the step's first kernel), queue,
name -- to see what runs beside what (side streams) and where a stream waits.
    rocpd_timeline.py results.db <name of the kernel that starts a step> [out.txt]"""
import sqlite3, sys


def main(path, first, out=None):
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if not starts:
        raise SystemExit("no kernel named like %r" % first)
    i0 = starts[-2] if len(starts) > 1 else starts[-1]         # the last complete step
    i1 = starts[-1] if len(starts) > 1 else len(rows)
    t0 = rows[i0][1]
    lines = ["start_us   end_us   dur_us  queue  name"]
    for name, s, e, qid in rows[i0:i1]:
        lines.append("%8.1f %8.1f %8.1f  %5s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qid, name[:90]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    return txt


if __name__ == "__main__":
    print(main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None))
