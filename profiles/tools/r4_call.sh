export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/r4_tests_full2.log 2>&1; tail -16 $O/r4_tests_full2.log
bash profiles/tools/final_profiles.sh r4 > $O/r4_final.log 2>&1; tail -3 $O/r4_final.log
timeout -k 10 300 python profiles/tools/prefill_probe.py 32 64 96 128 256 512 1024 2048 2>/dev/null | grep "^prompt" | cut -c1-175 > $O/r4_prefill_probe_final.log; cat $O/r4_prefill_probe_final.log
