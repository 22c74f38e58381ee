#!/bin/bash
# rocprofv3 kernel trace of the registry ops (verify_prefix, kv_append family) under the device-timed microbenchmark,
# summarised per (kernel, grid): bash profiles/tools/microbench_ops_profile.sh <tag>
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${1:-ops_prof}
mkdir -p $OUT
python3 $R/llm-inference-lab_amd/scripts/microbench_verify.py $OUT/microbench.json > $OUT/microbench.log 2>&1 || { tail -20 $OUT/microbench.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ops -- python3 $R/llm-inference-lab_amd/scripts/microbench_verify.py > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
python3 $R/profiles/tools/ops_summary.py $OUT/trace > $OUT/ops_kernels.md
cat $OUT/ops_kernels.md
