export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_hip_pipeline_gpu.py tests/test_hip_fwd0_select_gpu.py tests/test_hip_configs45_gpu.py tests/test_full_size_gpu.py tests/test_hip_sampling_gpu.py tests/test_hip_paged_kv_gpu.py -m gpu -q -x > $O/r4_tail_tests.log 2>&1; tail -6 $O/r4_tail_tests.log
for r in 1 2 3; do python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['acceptance_rate'])"; done
