# End-of-round measurement set (one GPU call): bench line, rocprofv3 kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE in their
# own runs), step breakdown.  bash profiles/tools/final_profiles.sh <tag>
tag=${1:-final}
export TMPDIR=/tmp
O=$PWD/gpurun_out
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { tail -20 $O/${tag}_bench.err; exit 1; }
echo "bench done"; tail -c 600 $O/${tag}_bench.json; echo
B="python3 bench.py --steps 30 --warmup 5 --cpu-baseline-steps 0 --no-probe"
P="python3 bench.py --steps 10 --warmup 2 --cpu-baseline-steps 0 --no-probe"   # counter passes serialise the kernels: fewer steps
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o t -- $B > $O/${tag}_stats.log 2>&1 || { tail -20 $O/${tag}_stats.log; exit 1; }
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -o t -- $P > $O/${tag}_fetch.log 2>&1 || { tail -20 $O/${tag}_fetch.log; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -o t -- $P > $O/${tag}_write.log 2>&1 || { tail -20 $O/${tag}_write.log; exit 1; }
echo "write done"
ls $O/${tag}_stats $O/${tag}_fetch | head
