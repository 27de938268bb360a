"""Steady-state kernel launches per training step: difference of two rocprofv3 kernel traces of bench.py that differ only in --steps, so that
one-time work (first-step allocations, optimizer-state initialisation, weight casts at construction) cancels.
usage: python tools/launch_count.py <results_a.db> <steps_a> <results_b.db> <steps_b>"""
import re, sqlite3, sys

def calls(path):
    cur = sqlite3.connect(path).cursor()
    return {n: c for n, c in cur.execute('select name, total_calls from top_kernels').fetchall()}

a, na, b, nb = calls(sys.argv[1]), int(sys.argv[2]), calls(sys.argv[3]), int(sys.argv[4])
# bench.py runs every step count twice: the timed pass and an untimed pass of the same length with per-launch events (roofline.achieved)
dn = 2 * (na - nb)
rows = []
for k in sorted(set(a) | set(b)):
    d = (a.get(k, 0) - b.get(k, 0)) / dn
    if d:
        rows.append((d, re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', k))[:120]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
ours = sum(r[0] for r in rows if not (r[1].startswith('at::') or 'rocclr' in r[1] or r[1].startswith('void at::')))
print(f'kernel launches per training step in steady state: {tot:.1f}  (library kernels {ours:.1f}, torch / runtime kernels {tot - ours:.1f})')
print(f'traces: {na} and {nb} steps of `python3 bench.py --steps K --warmup 2 --no-cpu-baseline --no-parity`, per-step = difference / {dn} (timed + untimed event pass)')
print(f'all launches of the traces: {sum(a.values())} and {sum(b.values())}\n')
for d, k in rows:
    print(f'{d:7.1f}  {k}')
