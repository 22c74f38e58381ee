export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=60 > $O/r4_tests_full4.log 2>&1; tail -70 $O/r4_tests_full4.log | cut -c1-150
