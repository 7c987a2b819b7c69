#!/bin/bash
# kernel-level A/B of the merge kernels (rocprof --stats; parse the kernel_stats.csv with a CSV reader: kernel names hold commas): BASE vs variant libraries
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in ${MAB:-BASE Q_MRG1 BASE Q_MRG1}; do
  lib=x-map_amd/_variants/libxmap_$n.so; [ "$n" = BASE ] && lib=x-map_amd/libxmap_hip.so
  rm -rf gpurun_out/mab_$n
  XMAP_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mab_$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra > /dev/null 2> gpurun_out/mab_$n.err
  echo $n $(grep -h "k_merge" gpurun_out/mab_$n/*/*kernel_stats.csv | awk -F, '{printf "%s %.3f ms  ", substr($1,8,22), $4/1e6}')
  rm -f gpurun_out/mab_$n/*/*kernel_trace.csv
done
