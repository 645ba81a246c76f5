# Dev (GPU box, DEV library): conv2a's inference forward on the plane-slab fetch (RGP_C2A_SLAB=1) vs the row-wise fetch (0),
# alternating processes on one box.   bash scripts/dev_ab_slab.sh <out file>
O=${1:-gpurun_out/r05/ab_slab.txt}
DEV=recurrent_gaze_prediction_amd/librgp_hip_dev.so
for r in 1 2 3 4; do for m in 1 0; do
RGP_C2A_SLAB=$m timeout -k 10 200 python scripts/dev_with_lib.py $DEV bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab $m  step %.3f ms  conv2a %.3f ms  %s' % (j['ms_per_step'], j['stage_ms_per_step']['conv2a'], j['roofline']['kernel'][:28]))"
done; done | tee $O
