export TMPDIR=/tmp
O=gpurun_out
hipcc -O3 --offload-arch=gfx950 profiles/tools/microbench/stream_floor.hip -o $O/stream_floor && timeout -k 10 300 $O/stream_floor > $O/r4_stream_floor.log 2>&1; cat $O/r4_stream_floor.log; rm -f $O/stream_floor
