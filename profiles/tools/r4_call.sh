export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_hip_prefill_gemm_gpu.py tests/test_hip_persist_gpu.py -q > $O/r4_tests_4.log 2>&1; tail -5 $O/r4_tests_4.log
timeout -k 10 400 python profiles/tools/prefill_probe.py 32 128 256 512 1024 2048 > $O/r4_prefill_probe_after.log 2>&1; echo rc=$?; grep "^prompt" $O/r4_prefill_probe_after.log
python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 > $O/r4_bench_os.json 2> $O/r4_bench_os.err; python -c "
import json;d=json.loads(open('gpurun_out/r4_bench_os.json').read().strip().splitlines()[-1]);print('bench', d['ms_per_step'],d['value'],d['step_roofline_frac'],d['roofline']['avg_launch_us'], d['roofline'].get('forward0'))"
