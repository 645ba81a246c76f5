# Dev (GPU box; needs recurrent_gaze_prediction_amd/librgp_hip_dev.so = a `make DEV=1` build): the 14 x 14 window-pair
# filter-gradient kernel (conv4a / conv4b, RGP_WGPATCH bits 3, 4) against wgrad_kernel: parity of the backward tests,
# an interleaved A/B of the fine-tune step, per-kernel times.   bash scripts/dev_ab_wgrad14.sh <outdir>
O=${1:-gpurun_out/r05/wg14}; mkdir -p $O
DEV=recurrent_gaze_prediction_amd/librgp_hip_dev.so
timeout -k 10 600 python -m pytest tests/test_c3d_backward_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -3 $O/t.log
for m in 7 31 7 31 7 31; do RGP_WGPATCH=$m timeout -k 10 200 python scripts/dev_with_lib.py $DEV bench.py --workload finetune --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m', j['ms_per_step'])"; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload finetune --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ft.json 2> $O/prof.err
grep -h "wgrad" $O/prof/*/*kernel_stats.csv | cut -d, -f1-4 | tee $O/wgrad_stats.txt
