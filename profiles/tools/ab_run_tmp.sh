for v in head new; do
  if [ $v = head ]; then export SPECDEC_HIP_LIB=_ab_head/libspecdec_hip.so; else unset SPECDEC_HIP_LIB; fi
  PROBE_WHICH=0,2 timeout -k 10 300 python profiles/tools/skinny_timeline.py 3b 20 40 2>&1 | grep "which=" | sed "s/^/$v /"
  PROBE_WHICH=0,2 timeout -k 10 300 python profiles/tools/skinny_timeline.py 8b 20 2>&1 | grep "which=" | sed "s/^/$v 8b /"
done
for rep in 1 2; do for v in head new; do
  if [ $v = head ]; then export SPECDEC_HIP_LIB=_ab_head/libspecdec_hip.so; else unset SPECDEC_HIP_LIB; fi
  python bench.py --batch 8 --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b8', round(d['ms_per_step'],4), end=' | ')"
  python bench.py --batch 4 --target llama-3-8b --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg4', round(d['ms_per_step'],4), end=' | ')"
done; echo; done
