# Dev (GPU box): fine-tune step (16 clips x T = 16 = 256 windows) of the in-tree library against another build, alternating
# processes on one box.   bash scripts/dev_ab_finetune.sh <other lib.so> [rounds] [out file]
OTHER=${1:-recurrent_gaze_prediction_amd/librgp_hip_prev.so}; R=${2:-3}; O=${3:-gpurun_out/r05/ab_finetune.txt}
for r in $(seq $R); do for lib in recurrent_gaze_prediction_amd/librgp_hip.so $OTHER; do
timeout -k 10 200 python scripts/dev_with_lib.py $lib bench.py --workload finetune --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']; print('$lib  step %.3f ms  conv2a %.3f conv3b %.3f conv4b %.3f' % (j['ms_per_step'], s['conv2a'], s['conv3b'], s['conv4b']))"
done; done | tee $O
