# Dev (GPU box): config 5's joint step (16 clips x T = 35 = 560 windows) and the cascade alone (forward / forward + backward)
# of the in-tree library against another build, alternating processes.   bash scripts/dev_ab_cfg5.sh <other lib.so> [rounds] [out]
OTHER=${1:-recurrent_gaze_prediction_amd/librgp_hip_prev.so}; R=${2:-3}; O=${3:-gpurun_out/r05/ab_cfg5.txt}
for r in $(seq $R); do for lib in recurrent_gaze_prediction_amd/librgp_hip.so $OTHER; do
timeout -k 10 300 python scripts/dev_with_lib.py $lib scripts/bench_config5.py --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib  config 5 joint step %.3f ms' % j['ms_per_step'])"
timeout -k 10 300 python scripts/dev_with_lib.py $lib scripts/dev_cascade_profile.py 2>/dev/null | tail -2 | sed "s|^|$lib  |"
done; done | tee $O
