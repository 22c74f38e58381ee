#!/bin/bash
# End-of-round measurement set, summarised ON the GPU box (raw traces are too large to copy back):
#   bash profiles/tools/round2_profiles.sh <tag> [part ...]     parts: headline b8 cfg4 secondary (default: all)
# Writes gpurun_out/<tag>/: bench.json, kernel_stats.csv, kernel_summary.md (+ PMC traffic), pmc_traffic.json (stamped with the
# library hash), step_breakdown.md, batch8_*.md, config4_*.md, secondary_configs.md
tag=${1:-r2_final}; shift
parts=${@:-headline b8 cfg4 secondary}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$tag
LIB=$R/llm-inference-lab_amd/lib/libspecdec_hip.so
mkdir -p $O
profile() {   # profile <name> <workload string> <bench flags...>
  name=$1; wl=$2; shift 2
  B="python3 bench.py $* --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
  P="python3 bench.py $* --steps 8 --warmup 2 --cpu-baseline-steps 0 --no-probe"   # counter passes serialise the kernels
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -o t -- $B > $O/${name}_stats.log 2>&1 || { tail -20 $O/${name}_stats.log; return 1; }
  echo "[$name] stats done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -o t -- $P > $O/${name}_fetch.log 2>&1 || { tail -20 $O/${name}_fetch.log; return 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -o t -- $P > $O/${name}_write.log 2>&1 || { tail -20 $O/${name}_write.log; return 1; }
  echo "[$name] pmc done"
  python3 profiles/summarize.py $O/${name}_stats $O/${name}_fetch $O/${name}_write $O/${name}_pmc_traffic.json $LIB "$wl" > $O/${name}_kernel_summary.md
  find $O/${name}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  find $O/${name}_stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 profiles/tools/step_breakdown.py {} > $O/${name}_step_breakdown.md
  grep '^{"metric"' $O/${name}_stats.log | tail -1 > $O/${name}_bench_under_profiler.json
  rm -rf $O/${name}_stats $O/${name}_fetch $O/${name}_write
  head -12 $O/${name}_kernel_summary.md
}
for part in $parts; do
  case $part in
    headline)
      python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
      echo "bench done"; tail -c 400 $O/bench.json; echo
      profile headline "llama-3.2-3b+llama-3.2-1b K=4 B=1" ;;
    b8) profile batch8 "llama-3.2-3b+llama-3.2-1b K=4 B=8" --batch 8 ;;
    cfg4) profile config4 "llama-3-8b+llama-3.2-1b K=4 B=4" --batch 4 --target llama-3-8b ;;
    secondary) bash profiles/tools/secondary_configs.sh > $O/secondary_configs.md 2>&1; cat $O/secondary_configs.md ;;
  esac
done
ls -la $O | head -40
