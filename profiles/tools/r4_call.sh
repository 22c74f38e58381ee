export TMPDIR=/tmp
O=gpurun_out
P="python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 100 --persist-only --no-timeline"
for rep in 1 2 3; do
  for v in main w0 w96 w192n4; do
    if [ $v = main ]; then unset SPECDEC_HIP_LIB; else export SPECDEC_HIP_LIB=_ab_$v/libspecdec_hip.so; fi
    echo -n "$v rep$rep: "; $P 2>/dev/null | grep "^persistent" | sed 's/persistent  persist_tokens=2 M=1: *//'
  done
done > $O/r4_ab_warm.log 2>&1
unset SPECDEC_HIP_LIB
cat $O/r4_ab_warm.log
python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 50 --persist-only > $O/r4_probe_warm_t1.log 2>&1
SPECDEC_HIP_LIB=_ab_diag/libspecdec_hip.so python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 50 --persist-only --diag > $O/r4_probe_warm_diag.log 2>&1
python -m pytest tests/test_hip_persist_gpu.py -q -x > $O/r4_tests_persist.log 2>&1; tail -5 $O/r4_tests_persist.log
