#!/bin/bash
# Dev: traffic beyond L2 (2 x FETCH_SIZE) and per-layer time of the C3D kernels of the current build: one FETCH_SIZE pass
# + one plain bench.  usage (on the GPU box): bash scripts/dev_fetch.sh [tag]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-cur}
O=gpurun_out/r03fetch_$T
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err || exit 1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > $O/time.json
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob('%s/fetch/**/*counter_collection.csv' % O, recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'conv' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-64:]].append(float(r['Counter_Value']))
d = json.loads(open('%s/time.json' % O).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items() if k.startswith('conv')})
for k, v in agg.items():
    print('   %-66s 2xFETCH %.2f GB per launch' % (k, 2 * sum(v) / len(v) * 1024 / 1e9))
PY
