for rep in 1 2 3 4; do for v in notaps taps; do
  if [ $v = taps ]; then export SPECDEC_PERSIST_TAPS=1; else unset SPECDEC_PERSIST_TAPS; fi
  python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), end=' | ')"
done; echo; done
