#!/bin/bash
# Build variants of csrc/persist.hip (compile-time switches) into _ab_<name>/libspecdec_hip.so for same-box A/B runs, in parallel:
#   profiles/tools/ab_persist.sh name "-DSD_P_WARM=96" [name2 "flags2" ...]
# (persist.hip is built at -Os, as llm-inference-lab_amd/build.py does: PERSIST_OPT overrides)
# then on the GPU box: SPECDEC_HIP_LIB=_ab_<name>/libspecdec_hip.so python profiles/tools/persist_probe.py ...
set -e
cd "$(dirname "$0")/../.."
python llm-inference-lab_amd/build.py > /dev/null
OBJS=$(ls llm-inference-lab_amd/csrc/.obj/*.o | grep -v "/persist\.")
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p _ab_$name
  ( hipcc ${PERSIST_OPT:--Os} -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Iinclude $flags -x hip -c ${PERSIST_SRC:-llm-inference-lab_amd/csrc/persist.hip} -o _ab_$name/persist.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS _ab_$name/persist.o -o _ab_$name/libspecdec_hip.so &&
    echo "built _ab_$name ($flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
