# Dev: SQ / LDS counter passes of the fine-tune step (separate --pmc runs), condensed by scripts/pmc_summary.py.
# Usage (GPU box, repo root): bash scripts/dev_pmc_finetune.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/pmc_ft}; mkdir -p $O
A="--workload finetune --batch 16 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py $A > $O/pmc_sq.json 2> $O/pmc_sq.err || echo "sq pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py $A > $O/pmc_lds.json 2> $O/pmc_lds.err || echo "lds pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $A > $O/pmc_fetch.json 2> $O/pmc_fetch.err || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $A > $O/pmc_write.json 2> $O/pmc_write.err || echo "write pass failed"
python3 scripts/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
