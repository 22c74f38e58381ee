export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 40 --warmup 5 --cpu-baseline-steps 0"
run() { echo "== $1"; for r in 1 2; do env $2 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['acceptance_rate'])"; done; }
{
run off "SPECDEC_NO_WARM_AHEAD=1"
run w31_2560 "SPECDEC_WARM_WGS=31 SPECDEC_WARM_KB=2560"
run w31_1024 "SPECDEC_WARM_WGS=31 SPECDEC_WARM_KB=1024"
run w31_512 "SPECDEC_WARM_WGS=31 SPECDEC_WARM_KB=512"
} > $O/r4_warm_ab2.log 2>&1
cat $O/r4_warm_ab2.log
export SPECDEC_WARM_WGS=31 SPECDEC_WARM_KB=2560
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/r4_warm_stats -o t -- python /root/repo/bench.py --steps 20 --warmup 3 --cpu-baseline-steps 0 > /root/repo/$O/r4_warm_prof.log 2>&1; cd /root/repo
python profiles/tools/step_breakdown.py $(find $O/r4_warm_stats -name "*kernel_trace.csv" | head -1) | head -14 | cut -c1-160
