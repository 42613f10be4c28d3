"""per-kernel averages of every counter in one rocprofv3 --pmc database, for the matrix kernels of the c3 step"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
KEYS = ("gemm", "mlp", "interp", "Cijk", "qinterp")
try:
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    print("# kernels columns:", cols)
except Exception as e:
    print("# no kernels view", e)
dur = {}
try:
    for r in db.execute("select name, avg(end-start), count(*) from kernels group by name").fetchall():
        dur[r[0]] = (r[1], r[2])
except Exception as e:
    print("# dur", e)
extra = {}
for cand in ("lds_size", "lds_block_size", "vgpr_count", "arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "scratch_size"):
    if cand in cols:
        for r in db.execute("select name, avg(%s) from kernels group by name" % cand).fetchall():
            extra.setdefault(r[0], {})[cand] = r[1]
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
per = {}
for k, c, v, n in rows:
    per.setdefault(k, {})[c] = v
for k in sorted(per):
    if not any(s in k for s in KEYS):
        continue
    d = dur.get(k, (0, 0))
    print("%-90s dur %.1f us x%d %s" % (k[:90], d[0] / 1e3, d[1], " ".join("%s=%g" % kv for kv in sorted(extra.get(k, {}).items()))))
    print("    " + "  ".join("%s=%.6g" % kv for kv in sorted(per[k].items())))
