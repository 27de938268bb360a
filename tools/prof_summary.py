"""Summarise a rocprofv3 results .db (kernel trace) into a short per-kernel table (name, calls, total, avg, %)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:110]
print(f'{"kernel":112s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}')
for n, c, t, a, p in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f'{short(n):112s} {c:6d} {t:12.1f} {a:10.2f} {p:6.2f}')
print('total_us', sum(r[2] for r in rows))
print('total_launches', sum(r[1] for r in rows), '(all kernels of the traced process, warm-up steps and one-time initialisation included)')
