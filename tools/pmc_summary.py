"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch."""
import csv, sys, collections, re, glob
path = sys.argv[1]
files = glob.glob(path + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', row.get('Kernel_Name', ''))[:60]
        cn = row.get('Counter_Name'); cv = float(row.get('Counter_Value', 0) or 0)
        a = acc[name][cn]; a[0] += cv; a[1] += 1
rows = sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))
for name, ctrs in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f'{name:62s} ' + ' '.join(f'{c}: mean {v[0]/max(v[1],1):.1f} over {v[1]} dispatches' for c, v in ctrs.items()))
