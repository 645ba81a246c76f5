# Dev (GPU box): config 4's training step replayed as HIP graphs (bench.py --workload train --graph) of the in-tree library
# against another build, alternating processes.   bash scripts/dev_ab_graph.sh <other lib.so> [rounds] [out file]
OTHER=${1:-recurrent_gaze_prediction_amd/librgp_hip_prev.so}; R=${2:-3}; O=${3:-gpurun_out/r05/ab_graph.txt}
for r in $(seq $R); do for lib in recurrent_gaze_prediction_amd/librgp_hip.so $OTHER; do
timeout -k 10 200 python scripts/dev_with_lib.py $lib bench.py --workload train --batch 8 --n-steps 35 --steps 300 --warmup 20 --graph --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib  graph replay B8xT35 %.4f ms' % j['ms_per_step'])"
done; done | tee $O
