# MFMA-busy evidence for the dominant kernels (north_star: "rocprof ... MFMA-busy counters"): counter passes of the bench,
# each group in its own run (no trace domains next to --pmc).  bash profiles/tools/mfma_counters.sh <tag>
tag=${1:-mfma}
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --list-avail > $O/${tag}_avail.txt 2>&1
grep -i "mfma" $O/${tag}_avail.txt | head -40
P="python3 bench.py --steps 10 --warmup 2 --cpu-baseline-steps 0 --no-probe"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_busy -o t -- $P > $O/${tag}_busy.log 2>&1 || tail -5 $O/${tag}_busy.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv -d $O/${tag}_mops -o t -- $P > $O/${tag}_mops.log 2>&1 || tail -5 $O/${tag}_mops.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/${tag}_act -o t -- $P > $O/${tag}_act.log 2>&1 || tail -5 $O/${tag}_act.log
ls $O/${tag}_busy $O/${tag}_mops $O/${tag}_act 2>&1 | head -12
