export TMPDIR=/tmp
O=gpurun_out
P="python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 100 --persist-only --no-timeline"
for r in 1 2 3; do
  for v in main lead0 lead3; do
    if [ $v = main ]; then unset SPECDEC_HIP_LIB; else export SPECDEC_HIP_LIB=_ab_$v/libspecdec_hip.so; fi
    echo "== $v run $r"; $P 2>&1 | grep -E "us|persist" | tail -3
  done
done > $O/r4_lead3_ab.log 2>&1
tail -40 $O/r4_lead3_ab.log
export SPECDEC_HIP_LIB=_ab_lead3/libspecdec_hip.so
python -m pytest tests/test_hip_persist_gpu.py -m gpu -q -x > $O/r4_lead3_tests.log 2>&1; tail -5 $O/r4_lead3_tests.log
