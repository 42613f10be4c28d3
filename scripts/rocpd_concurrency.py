"""Throughput mode under the kernel trace: for the densest window of a rocprofv3 rocpd database (many graphs in flight), per kernel
name -- launches, mean duration inside the window, and its share of the summed kernel time; plus the mean number of kernels
running at once.  Compared with the same kernels' durations in the eager (one batch in flight) trace this shows which
kernels stretch when the streams overlap.
    rocpd_concurrency.py throughput.db [eager.db] [window_ms]"""
import sqlite3, sys
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    return db.execute("select name, start, end from kernels order by start").fetchall()


def short(n):
    n = n.replace("ws3d::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:60]


def main(tp, eager=None, window_ms=200.0):
    rows = load(tp)
    # densest window: slide over starts, count kernels started inside
    import bisect
    starts = [r[1] for r in rows]
    w = int(window_ms * 1e6)
    marks = [r[1] for r in rows if "fps_bucket" in r[0]]      # one per batch: the window with the most batches
    best, t0 = -1, starts[0]
    for i, m in enumerate(marks):
        j = bisect.bisect_left(marks, m + w)
        if j - i > best:
            best, t0 = j - i, m
    t1 = t0 + w
    sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    tot = defaultdict(lambda: [0, 0.0])
    for n, s, e in sel:
        k = short(n); tot[k][0] += 1; tot[k][1] += (e - s)
    busy = sum(v[1] for v in tot.values())
    ref = {}
    if eager:
        acc = defaultdict(lambda: [0, 0.0])
        for n, s, e in load(eager):
            k = short(n); acc[k][0] += 1; acc[k][1] += (e - s)
        ref = {k: v[1] / v[0] for k, v in acc.items()}
    nfps = tot.get("fps_bucket_kernel", [0])[0]
    print("window %.0f ms, %d kernels, summed kernel time / window = %.1f kernels running on average; %d level-1 FPS launches -> %.3f ms per batch"
          % (window_ms, len(sel), busy / w, nfps, window_ms / max(nfps, 1)))
    print("%-62s %7s %10s %10s %8s %8s" % ("kernel", "calls", "mean us", "eager us", "stretch", "share"))
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
        r = ref.get(k)
        print("%-62s %7d %10.1f %10s %8s %7.1f%%" % (k, c, t / c / 1e3, "%.1f" % (r / 1e3) if r else "-", "%.2f" % (t / c / r) if r else "-", 100 * t / busy))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None, float(sys.argv[3]) if len(sys.argv) > 3 else 200.0)
