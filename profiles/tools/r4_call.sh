export TMPDIR=/tmp
O=gpurun_out
P="python profiles/tools/persist_probe.py --model 1b --tokens 1 --iters 100 --persist-only --no-timeline"
for rep in 1 2 3; do
  for v in main al6 al5 alb4 ilp memcl os o2; do
    if [ $v = main ]; then unset SPECDEC_HIP_LIB; else export SPECDEC_HIP_LIB=_ab_$v/libspecdec_hip.so; fi
    echo -n "$v rep$rep: "; $P 2>/dev/null | grep "^persistent" | sed 's/persistent  persist_tokens=2 M=1: *//'
  done
done > $O/r4_ab_flags.log 2>&1
unset SPECDEC_HIP_LIB
cat $O/r4_ab_flags.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r4_prefill_prof -o t -- python3 $GRAFT_REPO_ROOT/profiles/tools/prefill_probe.py 512 > $GRAFT_REPO_ROOT/$O/r4_prefill_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r4_prefill_prof/**/*kernel_stats.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:22]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
