export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_hip_pipeline_gpu.py -m gpu -q -x -k "step_tail or k_sweep" > $O/r4_tail_edge.log 2>&1; tail -12 $O/r4_tail_edge.log | cut -c1-200
