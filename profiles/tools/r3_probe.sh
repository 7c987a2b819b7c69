#!/bin/bash
# round-3 probe: box facts, the round-2 tree on the same box, kernel stats of the current tree (run from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1
(rocm-smi --showclocks --showmemuse --showpower --showperflevel 2>&1 | head -60; rocminfo 2>/dev/null | grep -E "Marketing|Compute Unit|Max Clock|Size:" | head -20; cat /sys/kernel/mm/transparent_hugepage/enabled) > gpurun_out/${T}_box.txt 2>&1
if [ -d _r02 ]; then
  (cd _r02 && timeout -k 10 200 python3 bench.py --steps 3 --warmup 2 --no-extra --no-cpu > ../gpurun_out/${T}_r02tree.json 2> ../gpurun_out/${T}_r02tree.err)
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err
rm -f gpurun_out/${T}_stats/*/*kernel_trace.csv
echo done
