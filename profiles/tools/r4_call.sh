export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/r4_tests_full3.log 2>&1; tail -14 $O/r4_tests_full3.log
for r in 1 2 3; do python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['acceptance_rate'])"; done
