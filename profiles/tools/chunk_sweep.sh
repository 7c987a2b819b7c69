#!/bin/bash
# sweep XMAP_CHUNK_DIV: paths bracket (enumeration + merges of the split starts) against the number of heavy rows
for d in 2048 4096 8192 16384; do
  XMAP_CHUNK_DIV=$d timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/r4k_div_$d.json 2> gpurun_out/r4k_div_$d.err || true
  python3 - $d gpurun_out/r4k_div_$d.json <<'P'
import sys, json
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("div", sys.argv[1], "paths %.2f" % d["kernel_ms"]["paths"], "B %.2f" % d["stage_ms"]["B_extend"], "heavy_rows", d["accumulator_rows"]["heavy_rows"])
except Exception as e:
    print(sys.argv[1], "failed", e)
P
done
