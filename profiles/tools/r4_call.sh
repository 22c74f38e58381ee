export TMPDIR=/tmp
O=gpurun_out
python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 100 > $O/r4_probe_base_t1.log 2>&1 &&
python profiles/tools/persist_probe.py --model 1b --tokens 2 --iters 100 --persist-only > $O/r4_probe_base_t2.log 2>&1 &&
SPECDEC_HIP_LIB=_ab_diag/libspecdec_hip.so python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 50 --persist-only --diag > $O/r4_probe_diag_t1.log 2>&1 &&
SPECDEC_HIP_LIB=_ab_diag/libspecdec_hip.so python profiles/tools/persist_probe.py --model 1b --tokens 2 --iters 50 --persist-only --diag > $O/r4_probe_diag_t2.log 2>&1 &&
python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0 > $O/r4_bench_base.json 2> $O/r4_bench_base.err &&
bash profiles/tools/mfma_counters.sh r4_mfma > $O/r4_mfma.log 2>&1
echo rc=$?
tail -3 $O/r4_probe_base_t1.log; tail -c 400 $O/r4_bench_base.json
