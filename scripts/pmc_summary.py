"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, mean of each counter."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    n = max(len(v) for v in cs.values())
    print(k, 'dispatches', n)
    for c, v in sorted(cs.items()):
        print('    %-28s mean %.4g' % (c, sum(v) / len(v)))
