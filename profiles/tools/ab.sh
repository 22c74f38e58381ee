# A/B of two builds of the library on the headline bench: bash profiles/tools/ab.sh <tag> [bench flags]
# (lib/libspecdec_hip_base.so = the build to compare against, copied by hand before rebuilding)
tag=$1; shift
B="python bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe $*"
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export SPECDEC_HIP_LIB=$PWD/llm-inference-lab_amd/lib/libspecdec_hip_base.so; else unset SPECDEC_HIP_LIB; fi
    $B > gpurun_out/ab_${tag}_$v.json 2> gpurun_out/ab_${tag}_$v.err || exit 1
    python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_${tag}_$v.json').read().strip().splitlines()[-1])
print('$tag $v rep$rep', round(d['ms_per_step'],4), 'ms/step', round(d['value'],1), 'tok/s', flush=True)"
  done
done
