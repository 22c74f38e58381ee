export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 300 python profiles/tools/cu_mask_overlap.py --iters 30 > $O/r4_cu_mask_overlap.log 2>&1; echo rc=$?; tail -8 $O/r4_cu_mask_overlap.log
python -m pytest tests/test_hip_persist_gpu.py tests/test_full_size_gpu.py -q > $O/r4_tests_2.log 2>&1; tail -8 $O/r4_tests_2.log
