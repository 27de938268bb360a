"""Dev: for a kernel-name pattern, list which kernels run right before it (rocprofv3 results .db, kernel trace order)."""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
kt = [t for t in tabs if 'kernel_dispatch' in t.lower() or t == 'kernels']
print('tables:', [t for t in tabs if 'kernel' in t.lower()][:10])
t = 'kernels' if 'kernels' in tabs else kt[0]
cols = [r[1] for r in cur.execute(f'pragma table_info({t})').fetchall()]
print(t, cols[:25])
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
start_col = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
gcols = [c for c in cols if c.startswith('grid') or c.startswith('workgroup')]
rows = cur.execute(f'select {name_col}, {start_col}, {", ".join(gcols) if gcols else "0"} from {t} order by {start_col}').fetchall()
pat = sys.argv[2]
prev = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev[(rows[i - 1][0][:70] if i else '-', (rows[i + 1][0][:50] if i + 1 < len(rows) else '-'))] += 1
        sizes[tuple(r[2:])] += 1
for k, v in prev.most_common(25): print(v, k)
print('grid/wg sizes:', sizes.most_common(12))
