"""Per-kernel time and idle gaps of one implicit step from a rocprofv3 --kernel-trace database.
usage: python tools/trace_step.py results.db [kernels_per_step]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
rows = list(cur.execute(f"select kernel_id, start, end from {kd} order by start"))


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void kry::", "").replace("(anonymous namespace)::", "")
    return n


# one step = from one k_pack_wrap to the next
starts = [i for i, r in enumerate(rows) if "k_pack_wrap" in names[r[0]]]
if len(starts) < 3:
    sys.exit("no steps found")
a, b = starts[-2], starts[-1]
step = rows[a:b]
busy = collections.defaultdict(lambda: [0, 0.0])
gap_after = collections.defaultdict(lambda: [0, 0.0])
for i, (k, s, e) in enumerate(step):
    nm = re.sub(r"<.*", "", short(names[k]))
    busy[nm][0] += 1
    busy[nm][1] += (e - s) / 1e3
    if i + 1 < len(step):
        gap_after[nm][0] += 1
        gap_after[nm][1] += max(0, step[i + 1][1] - e) / 1e3
total = (rows[b][1] - rows[a][1]) / 1e3
print("step: %d kernels, %.1f us from first start to next step's start" % (len(step), total))
print("%-28s %6s %10s %8s %12s" % ("kernel", "calls", "busy us", "avg us", "gap after us"))
tb = tg = 0.0
for nm, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    g = gap_after[nm][1]
    tb += t
    tg += g
    print("%-28s %6d %10.1f %8.2f %12.1f" % (nm, c, t, t / c, g))
print("busy %.1f us, gaps %.1f us" % (tb, tg))
