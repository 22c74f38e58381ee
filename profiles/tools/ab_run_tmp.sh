for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export SPECDEC_HIP_LIB=_ab_oldattn/libspecdec_hip.so; else unset SPECDEC_HIP_LIB; fi
  python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), end=' | ')"
done; echo; done
