set -e
B="python bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
run() { name=$1; shift; env "$@" $B > gpurun_out/knob_$name.json 2> gpurun_out/knob_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/knob_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["value"], flush=True)
PY
}
run base X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run fgs1 ROC_USE_FGS_KERNARG=1
run fgs0 ROC_USE_FGS_KERNARG=0
