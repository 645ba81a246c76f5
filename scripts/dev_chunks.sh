for c in 16 24 32 48 64 128 256 1024; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --c3d-chunk $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['stage_ms_per_step']
print('chunk',$c,'ms/step',d['ms_per_step'],' '.join('%s=%.2f'%(k,s[k]) for k in ('conv1a','conv2a','conv3a','conv3b','conv4a','conv4b','conv5a','conv5b')))
"
done
