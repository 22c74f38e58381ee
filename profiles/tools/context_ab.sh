for v in base new; do
  if [ $v = base ]; then export SPECDEC_HIP_LIB=$PWD/llm-inference-lab_amd/lib/libspecdec_hip_base.so; else unset SPECDEC_HIP_LIB; fi
  echo "== $v"; timeout -k 10 400 python profiles/tools/context_scaling.py 2>&1 | grep "^context"
done
unset SPECDEC_HIP_LIB
python -m pytest tests/test_hip_kernels_gpu.py tests/test_hip_forward_gpu.py -x -q -k "attention or context or split or forward" 2>&1 | tail -2
