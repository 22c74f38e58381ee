export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_hip_pipeline_gpu.py tests/test_hip_sampling_gpu.py tests/test_hip_wrappers_gpu.py -m gpu -q -x > $O/r4_sampled_tests.log 2>&1; tail -25 $O/r4_sampled_tests.log
