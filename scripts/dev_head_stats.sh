#!/bin/bash
# Dev: rocprofv3 kernel stats of the head-only forward (B64 x T16); usage: dev_head_stats.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/head_$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload head --steps 50 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print('%8.1f us/step %5.1f calls  avg %7.1f us  %s' % (float(r['TotalDurationNs']) / 55e3, float(r['Calls']) / 55.0, float(r['AverageNs']) / 1e3, r['Name'][:150]))
PY
find $O -name "*kernel_trace.csv" -size +4M -delete
