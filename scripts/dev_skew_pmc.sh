#!/bin/bash
# Dev (DEV=1 build on the box): traffic beyond L2 and time of the C3D patch kernels with the CUs of an XCD started out of
# phase (RGP_CP_ABLATE = skew << 8), FETCH_SIZE pass + plain timing pass per setting.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -C recurrent_gaze_prediction_amd/csrc clean > /dev/null
make -C recurrent_gaze_prediction_amd/csrc DEV=1 -j16 > gpurun_out/r03_devbuild.log 2>&1 || exit 1
O=gpurun_out/r03skew
mkdir -p $O
for skew in 0 8; do
  export RGP_CP_ABLATE=$((skew << 8))
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$skew -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch_$skew.json 2> $O/fetch_$skew.err || exit 1
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null > $O/time_$skew.json
  python3 - "$O" "$skew" <<'PY'
import csv, glob, json, sys, collections
O, skew = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob('%s/fetch_%s/**/*counter_collection.csv' % (O, skew), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'conv_patch' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0][-60:]].append(float(r['Counter_Value']))
d = json.loads(open('%s/time_%s.json' % (O, skew)).read().strip().splitlines()[-1])['stage_ms_per_step']
print('skew', skew, {k: round(v, 2) for k, v in d.items() if k.startswith('conv')})
for k, v in agg.items():
    print('   %-62s 2xFETCH %.2f GB per launch' % (k, 2 * sum(v) / len(v) * 1024 / 1e9))
PY
done
