export TMPDIR=/tmp
O=gpurun_out
python bench.py --steps 5000 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>$O/r4_long.err | tail -1 > $O/r4_long.json; python -c "
import json; d=json.load(open('$O/r4_long.json')); print({k:d[k] for k in ('ms_per_step','value','steps','resyncs','acceptance_rate','step_roofline_frac')})"
tail -3 $O/r4_long.err
