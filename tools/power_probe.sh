#!/bin/bash
# Is the headline step power / clock limited?  Step time for short and long timed regions, with rocm-smi samples of the
# shader clock and the socket power taken while the long one runs.  -> gpurun_out/power_probe.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
mkdir -p gpurun_out; out=gpurun_out/power_probe.txt; : > $out
for k in 10 40 100 400; do
  if [ $k = 400 ]; then
    ( for i in $(seq 1 12); do sleep 0.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; done ) > gpurun_out/smi_samples.txt &
    SMI=$!
  fi
  python bench.py $F --steps $k --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('steps=$k', j['value'], 'views/s', j['ms_per_step'], 'ms  hip_kernel_ms', j.get('hip_kernel_ms_per_step'))" >> $out
done
wait $SMI 2>/dev/null
echo "--- rocm-smi samples during the 400-step run" >> $out
cat gpurun_out/smi_samples.txt >> $out
cat $out
