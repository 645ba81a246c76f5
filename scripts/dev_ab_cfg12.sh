# Dev (GPU box): configs 1 (frame-wise ShallowNet) and 2 (fc-GRU) forward / training step of the in-tree library against another
# build, alternating processes.   bash scripts/dev_ab_cfg12.sh <other lib.so> [rounds] [out]
OTHER=${1:-recurrent_gaze_prediction_amd/librgp_hip_prev.so}; R=${2:-3}; O=${3:-gpurun_out/r05/ab_cfg12.txt}
for r in $(seq $R); do for lib in recurrent_gaze_prediction_amd/librgp_hip.so $OTHER; do
timeout -k 10 200 python scripts/dev_with_lib.py $lib scripts/dev_cfg1_time.py 2>/dev/null | tail -1 | sed "s|^|$lib  |"
timeout -k 10 200 python scripts/dev_with_lib.py $lib scripts/dev_cfg2_time.py 2>/dev/null | tail -1 | sed "s|^|$lib  |"
done; done | tee $O
