#!/bin/bash
# Dev: rocprofv3 kernel stats of the training-side workloads (fine-tune B16xT16, config-4 step B8xT35, config 2 fc-GRU)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03dev
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ft -- python3 bench.py --workload finetune --batch 16 --n-steps 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ft.json 2> $O/stats_ft.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_t35 -- python3 bench.py --workload train --batch 8 --n-steps 35 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_t35.json 2> $O/stats_t35.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 scripts/dev_cfg2_profile.py > $O/cfg2.txt 2> $O/stats_cfg2.err || exit 1
for d in ft t35 cfg2; do f=$(find $O/stats_$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -28 $f | cut -c1-200; done
find $O -name "*kernel_trace.csv" -size +4M -delete
