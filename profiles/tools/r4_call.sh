export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=5 > $O/r4_tests_full5.log 2>&1; tail -10 $O/r4_tests_full5.log | cut -c1-150
bash profiles/tools/final_profiles.sh r4 > $O/r4_final.log 2>&1; tail -3 $O/r4_final.log
for r in 1 2; do python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic'))"; done
