#!/bin/bash
# Attribution of the stage-A pair kernels: builds libxmap_hip.so variants with one memory stream removed each
# (-DEXP_*: switches in csrc/stage_a2.hip, results are WRONG by construction -- timing only) and, on the GPU box,
# times the pair phase of every variant with bench.py's own HIP-event brackets.
#   here:        profiles/tools/a_variants.sh build  "NONRM NOUAVG NOCOO"      (FILE=stage_b: flags of csrc/stage_b.hip, -DEXP_x)
#   on the box:  profiles/tools/a_variants.sh run TAG "BASE NONRM NOCNT NOUAVG NOCOO"
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
V=$ROOT/x-map_amd/_variants
C=$ROOT/x-map_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $V
  make -C $C -j8 >/dev/null
  for n in $2; do
    flags=""; for f in ${n//+/ }; do if [ "$f" = ATRACE ]; then flags="$flags -DA_TRACE"; elif [ "$f" = P_TRACE ]; then flags="$flags -DP_TRACE -DQ_PIPE"; elif [[ "$f" == Q_* || "$f" == P_WAVES=* ]]; then flags="$flags -D$f"; elif [[ "$f" == *=* ]]; then flags="$flags -DEXP_$f"; else flags="$flags -DEXP_$f"; fi; done
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $flags -c $C/${FILE:-stage_a2}.hip -o $V/a2_$n.o &
  done
  wait
  for n in $2; do
    objs=$(ls $C/_build/*.o | grep -v ${FILE:-stage_a2}.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libxmap_$n.so $objs $V/a2_$n.o
    rm -f $V/a2_$n.o
  done
  ls -la $V
else
  T=$2
  cd $ROOT
  for n in $3; do
    lib=$V/libxmap_$n.so; [ "$n" = BASE ] && lib=$ROOT/x-map_amd/libxmap_hip.so
    env_extra=""; [[ "$n" == *Q_STORE* ]] && env_extra="XMAP_ABL_ROW_ENTRIES=${ABL_ROW_ENTRIES:-830000}"
    env $env_extra XMAP_HIP_LIB=$lib timeout -k 10 ${VAR_TIMEOUT:-200} python3 bench.py --steps ${VAR_STEPS:-10} --warmup ${VAR_WARMUP:-3} --no-cpu --no-extra > gpurun_out/${T}_$n.json 2> gpurun_out/${T}_$n.err || true
    python3 - $n gpurun_out/${T}_$n.json <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["kernel_ms"]
    print(sys.argv[1], "stage_a %.3f" % d["stage_ms"]["A_item_sim"], " ".join("%s %.3f" % (x, k[x]) for x in ("layout3", "tri_plan", "pair_tri", "mir_count", "scatter", "paths", "knn_classify", "reverse", "mid_build") if x in k))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
  done
fi
