"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) into the same
table `rocprofv3 --stats` prints: per-kernel calls / total / average / min / max / percent."""
import sqlite3, sys

def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, t, a, mn, mx, 100.0 * t / tot))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    return txt

if __name__ == "__main__":
    print(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
