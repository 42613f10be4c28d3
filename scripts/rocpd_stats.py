"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) into the same
table `rocprofv3 --stats` prints: per-kernel calls / total / average / min / max / percent.

    rocpd_stats.py results.db [out.csv] [--last-ms T]

--last-ms T keeps only the launches that started in the final T milliseconds of the trace
(steady state of a training run: the first iterations are MIOpen's solver search)."""
import sqlite3, sys


def main(path, out=None, last_ms=None):
    db = sqlite3.connect(path)
    where = ""
    if last_ms is not None:
        t_end = db.execute("select max(end) from kernels").fetchone()[0]
        where = "where start >= %d" % (t_end - int(last_ms * 1e6))
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels %s group by name order by 3 desc" % where).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, t, a, mn, mx, 100.0 * t / tot))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    return txt


if __name__ == "__main__":
    argv = sys.argv[1:]
    last = None
    if "--last-ms" in argv:
        i = argv.index("--last-ms")
        last = float(argv[i + 1])
        del argv[i:i + 2]
    print(main(argv[0], argv[1] if len(argv) > 1 else None, last))
