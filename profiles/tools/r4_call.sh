export TMPDIR=/tmp
O=gpurun_out
bash profiles/tools/final_profiles.sh r4 > $O/r4_final.log 2>&1; tail -3 $O/r4_final.log
bash profiles/tools/secondary_configs.sh > $O/r4_secondary.log 2>&1; tail -16 $O/r4_secondary.log | cut -c1-200
