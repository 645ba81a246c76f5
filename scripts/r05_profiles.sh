#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root): kernel stats of the driver-style bench and of the
# head / train / fine-tune workloads, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ sets) of a short e2e run.
# Output under gpurun_out/r05/; scripts/pmc_summary.py condenses the PMC passes (-> profiles/r05_pmc_summary.json).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_e2e.json 2> $O/bench_e2e.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_e2e -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_e2e_under_rocprof.json 2> $O/stats_e2e.err || exit 1
echo "stats e2e done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_head -- python3 bench.py --workload head --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_head.json 2> $O/stats_head.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_B64_T16 -- python3 bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_train_B64_T16.json 2> $O/stats_train.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_B8_T35 -- python3 bench.py --workload train --batch 8 --n-steps 35 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_train_B8_T35.json 2> $O/stats_train35.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_finetune_B16_T16 -- python3 bench.py --workload finetune --batch 16 --n-steps 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_finetune_B16_T16.json 2> $O/stats_ft.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2_fcgru -- python3 scripts/dev_cfg2_profile.py train > $O/cfg2.txt 2> $O/stats_cfg2.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- python3 scripts/bench_config5.py > $O/bench_cfg5.json 2> $O/stats_cfg5.err || echo "cfg5 stats failed"
echo "stats others done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.json 2> $O/pmc_sq.err || echo "sq pmc pass failed (counter set)"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lds.json 2> $O/pmc_lds.err || echo "lds pmc pass failed (counter set)"
echo "pmc done"
python3 scripts/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
# the raw per-dispatch traces are large: keep the stats / summaries
find $O -name "*kernel_trace.csv" -size +4M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
ls $O
# round 4-5 extras: the data-parallel probes at N = 1 (no collective runs; the keys the driver's N > 1 command will carry)
python3 bench.py --steps 5 --dp-train-probe on --dp-finetune-probe on --no-cpu-baseline > $O/bench_probes.json 2> $O/bench_probes.err || echo "probes failed"
python3 scripts/bench_configs.py > $O/configs.json 2> $O/configs.err || echo "configs failed"
ls $O
