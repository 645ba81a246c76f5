# Dev (make DEV=1 build of rgp_c3d_bwd.o): parity of the patch filter-gradient kernels on all three layers, an
# interleaved A/B of the fine-tune step against wgrad_kernel, and their per-kernel times.  Usage (GPU box): bash scripts/dev_wgpatch_ab.sh <outdir>
O=${1:-gpurun_out/wgp}; mkdir -p $O
RGP_WGPATCH=7 timeout -k 10 300 python -m pytest tests/test_c3d_backward_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -3 $O/t.log
for m in 0 7 1 0 7 1; do RGP_WGPATCH=$m timeout -k 10 120 python bench.py --workload finetune --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m', j['ms_per_step'])"; done
export RGP_WGPATCH=7
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload finetune --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/prof.err
grep wgrad $O/prof/*/*kernel_stats.csv | cut -d, -f1-4
