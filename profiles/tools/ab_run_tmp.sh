for rep in 1 2 3; do for v in 8 4 6 10 12; do
  if [ $v = 8 ]; then unset SPECDEC_HIP_LIB; else export SPECDEC_HIP_LIB=_ab_kpre$v/libspecdec_hip.so; fi
  python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kpre$v', round(d['ms_per_step'],4), end=' | ')"
done; echo; done
