#!/bin/bash
# Copy the summaries of scripts/r0N_profiles.sh (gpurun_out/r0N/) into profiles/ (tracked).  usage: collect_profiles.sh [round tag]
R=${1:-r04}
O=gpurun_out/$R
st() { ls -t $(find $O/$1 -name "*kernel_stats.csv") | head -1; }   # the newest run (gpurun merges into an existing directory)
cp $O/bench_e2e.json profiles/${R}_bench_e2e.json
cp $O/bench_e2e_under_rocprof.json profiles/${R}_bench_e2e_under_rocprof.json
cp "$(st stats_e2e)" profiles/${R}_e2e_kernel_stats.csv
cp "$(st stats_head)" profiles/${R}_head_kernel_stats.csv
cp "$(st stats_train_B64_T16)" profiles/${R}_train_B64_T16_kernel_stats.csv
cp "$(st stats_train_B8_T35)" profiles/${R}_train_B8_T35_kernel_stats.csv
cp "$(st stats_finetune_B16_T16)" profiles/${R}_finetune_B16_T16_kernel_stats.csv
cp "$(st stats_cfg2_fcgru)" profiles/${R}_cfg2_fcgru_train_kernel_stats.csv
cp "$(st stats_cfg5)" profiles/${R}_cfg5_kernel_stats.csv
cp $O/pmc_summary.json profiles/${R}_pmc_summary.json
for f in bench_head bench_train_B64_T16 bench_train_B8_T35 bench_finetune_B16_T16 bench_cfg5; do cp $O/$f.json profiles/${R}_$f.json; done
[ -f $O/configs.json ] && cp $O/configs.json profiles/${R}_configs.json
[ -f $O/bench_probes.json ] && cp $O/bench_probes.json profiles/${R}_bench_probes.json
ls -la profiles | grep ${R}_
