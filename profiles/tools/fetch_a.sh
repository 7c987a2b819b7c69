#!/bin/bash
# usage: fetch_a.sh TAG "BASE NONRM NOUAVG ..."  (GPU box, repo root): FETCH_SIZE / WRITE_SIZE of the stage-A kernels per library
# variant (profiles/tools/a_variants.sh), one pass of stage A each -> gpurun_out/TAG_fetch_variants.txt
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in $2; do
  lib=$GRAFT_REPO_ROOT/x-map_amd/_variants/libxmap_$n.so; [ "$n" = BASE ] && lib=$GRAFT_REPO_ROOT/x-map_amd/libxmap_hip.so
  export XMAP_HIP_LIB=$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${T}_f_${n}_${c} -- python3 profiles/tools/run_a.py 1 > gpurun_out/${T}_f.txt 2> gpurun_out/${T}_f.err || exit 1
    python3 - $n $c gpurun_out/${T}_f_${n}_${c} <<'PY'
import csv, glob, collections, sys, re
tot = collections.defaultdict(float)
for fn in glob.glob(sys.argv[3] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xmap::", "").replace("xmap::", "")
        tot[k] += float(r["Counter_Value"])
pair = sum(v for k, v in tot.items() if k.startswith("k_pair_tri") or k.startswith("k_pair_heavy") or k.startswith("k_heavy_merge"))
f = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0      # KB; the guide's gfx950 correction doubles FETCH_SIZE
print(sys.argv[1], sys.argv[2], "pair kernels %.3f GB (guide's formula; x1: %.3f GB);" % (pair * 1024 * f / 1e9, pair * 1024 / 1e9),
      " ".join("%s %.3f" % (k[:28], v * 1024 * f / 1e9) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]))
PY
    rm -rf gpurun_out/${T}_f_${n}_${c}
  done
done
