#!/bin/bash
# Variants of ONE csrc file under other compiler flags, linked against the current objects of the others, for same-box A/B runs:
#   profiles/tools/ab_file.sh gemv.hip name "-O2" [name2 "flags2" ...]   -> _ab_<name>/libspecdec_hip.so
set -e
cd "$(dirname "$0")/../.."
python llm-inference-lab_amd/build.py > /dev/null
file=$1; shift
stem=${file%.*}
OBJS=$(ls llm-inference-lab_amd/csrc/.obj/*.o | grep -v "/$stem\.")
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p _ab_$name
  ( hipcc -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Iinclude $flags -x hip -c llm-inference-lab_amd/csrc/$file -o _ab_$name/$stem.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS _ab_$name/$stem.o -ldl -o _ab_$name/libspecdec_hip.so &&
    echo "built _ab_$name ($file: $flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
