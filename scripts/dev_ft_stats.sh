#!/bin/bash
# Dev: rocprofv3 kernel stats of the fine-tune step (B16 x T16 = 256 windows); top kernels per step on stdout.
# usage: dev_ft_stats.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ft_$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload finetune --batch 16 --n-steps 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ft.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total %.2f ms per step (13 steps profiled)' % (tot / 13e6))
for r in rows[:24]:
    print('%8.3f ms/step %6.1f calls/step  %s' % (float(r['TotalDurationNs']) / 13e6, float(r['Calls']) / 13.0, r['Name'][:110]))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench_ft.json
find $O -name "*kernel_trace.csv" -size +4M -delete
