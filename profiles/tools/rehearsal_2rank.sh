# 2-rank rehearsal of the multi-GPU bench path on a one-GPU box (gloo statistics exchange, both ranks on cuda:0, persistent launches off): bash profiles/tools/rehearsal_2rank.sh
export TMPDIR=/tmp
O=gpurun_out
SPECDEC_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --cpu-baseline-steps 0 > $O/r4_rehearsal2.log 2>&1; tail -1 $O/r4_rehearsal2.log | cut -c1-900
