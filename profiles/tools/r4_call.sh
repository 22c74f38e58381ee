export TMPDIR=/tmp
O=gpurun_out
hipcc -O3 --offload-arch=gfx950 profiles/tools/microbench/boundary.hip -o $O/boundary && timeout -k 10 300 $O/boundary > $O/r4_boundary.log 2>&1; cat $O/r4_boundary.log; rm -f $O/boundary
