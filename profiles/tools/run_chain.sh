# chained launches (gemv_chain.hip): parity test, in-kernel timelines, and the step with each pair chained
set -e
timeout -k 10 300 python -m pytest tests/test_hip_chain_gpu.py -x -q > gpurun_out/chain_test.log 2>&1 || { tail -30 gpurun_out/chain_test.log; exit 1; }
tail -1 gpurun_out/chain_test.log
SPECDEC_CHAIN_PAIRS=3 SPECDEC_GEMV_TIMELINE=1 timeout -k 10 300 python profiles/tools/probe_chain.py > gpurun_out/probe_chain_tl2.log 2>&1
grep "chained" gpurun_out/probe_chain_tl2.log
B="python bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
for v in "CHAIN_PAIRS=0" "CHAIN_PAIRS=1" "CHAIN_PAIRS=3" "CHAIN_PAIRS=2"; do
  env SPECDEC_$v timeout -k 10 200 $B > gpurun_out/chain_ab.json 2> gpurun_out/chain_ab.err || { tail -20 gpurun_out/chain_ab.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/chain_ab.json').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],4), 'ms/step', round(d['value'],1), 'tok/s', d.get('chained_launches'), flush=True)"
done
