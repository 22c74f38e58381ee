# Kernel stats + HBM traffic of a batched step (default: BASELINE config 3 shape, 3B + 1B, K=4, 8 rows per GPU; FLAGS="--batch 4 --target llama-3-8b" = config 4 per GPU).
tag=${1:-b8}
export TMPDIR=/tmp
O=$PWD/gpurun_out
FLAGS=${FLAGS:---batch 8}
B="python3 bench.py $FLAGS --steps 20 --warmup 3 --cpu-baseline-steps 0 --no-probe"
P="python3 bench.py $FLAGS --steps 6 --warmup 2 --cpu-baseline-steps 0 --no-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o t -- $B > $O/${tag}_stats.log 2>&1 || { tail -20 $O/${tag}_stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -o t -- $P > $O/${tag}_fetch.log 2>&1 || { tail -20 $O/${tag}_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -o t -- $P > $O/${tag}_write.log 2>&1 || { tail -20 $O/${tag}_write.log; exit 1; }
grep '^{"metric"' $O/${tag}_stats.log | tail -c 700
