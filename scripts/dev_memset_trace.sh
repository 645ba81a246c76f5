# Dev (GPU box): which memsets does config 5's joint step issue, and how long do they take?  Per-dispatch kernel trace,
# then the fill kernels with their grid sizes and durations (one step's worth printed).
O=gpurun_out/r05/memset_trace; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 scripts/bench_config5.py > $O/bench.json 2> $O/err.txt || echo "trace failed"
python3 - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/r05/memset_trace/t/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fills = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), r['Workgroup_Size'] if 'Workgroup_Size' in r else r.get('Workgroup_Size_X'), i) for i, r in enumerate(rows) if 'fillBuffer' in r['Kernel_Name']]
n = len(fills)
print('fill dispatches', n, 'columns', list(rows[0].keys()))
agg = collections.Counter(); cnt = collections.Counter()
for d, g, w, i in fills: agg[g] += d; cnt[g] += 1
for g, t in agg.most_common(12): print('grid', g, 'calls', cnt[g], 'total us', t / 1e3, 'avg us', t / 1e3 / cnt[g])
# what runs right after the 6 largest
big = sorted(fills, reverse=True)[:8]
for d, g, w, i in big:
    print(d / 1e3, 'us grid', g, 'prev:', rows[i - 1]['Kernel_Name'][:60], '| next:', rows[i + 1]['Kernel_Name'][:60])
P
