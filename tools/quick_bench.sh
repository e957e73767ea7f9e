#!/bin/bash
# run on the GPU box: MLP tests + full-step bench, top kernels
python -m pytest tests/test_mlp_gpu.py -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-codec --no-raster-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('value',d['value'],'ms',d['ms_per_step'],'hip',d['hip_kernel_ms_per_step'])
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['total_ms'])[:8]: print(n, v['avg_us'], v['launches'])"
