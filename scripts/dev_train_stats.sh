#!/bin/bash
# Dev: rocprofv3 kernel stats of the head training step; usage: dev_train_stats.sh <tag> <batch> <n_steps>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/train_$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload train --batch $2 --n-steps $3 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total %.3f ms per step (35 steps profiled)' % (tot / 35e6))
for r in rows[:32]:
    print('%8.1f us/step %5.1f calls  avg %7.1f us  %s' % (float(r['TotalDurationNs']) / 35e3, float(r['Calls']) / 35.0, float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.json
find $O -name "*kernel_trace.csv" -size +4M -delete
