B="python bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export SPECDEC_ONE_STREAM=1; else unset SPECDEC_ONE_STREAM; fi
  $B > gpurun_out/os_$v.json 2> gpurun_out/os_$v.err
  python -c "
import json
d=json.loads(open('gpurun_out/os_$v.json').read().strip().splitlines()[-1])
print('ONE_STREAM=$v', round(d['ms_per_step'],4), round(d['value'],1), flush=True)"
done; done
