python -m pytest tests/test_hip_pipeline_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do for v in sel nosel; do
  if [ $v = nosel ]; then export SPECDEC_NO_FWD0_SELECT=1; else unset SPECDEC_NO_FWD0_SELECT; fi
  python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['tokens_per_step'],3), end=' | ')"
done; echo; done
