#!/bin/bash
# usage: profile_round.sh TAG   (on the GPU box, from the repo root) -- the four runs behind profiles/<TAG>_c2_*
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || exit 1
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err || exit 1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra > gpurun_out/${T}_fetch.json 2> gpurun_out/${T}_fetch.err || exit 1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra > gpurun_out/${T}_write.json 2> gpurun_out/${T}_write.err || exit 1
echo write done
# SQ counters of the path kernel (two passes of 8 counters), summed over its launches -> one small JSON
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/${T}_sq1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra > gpurun_out/${T}_sq1.json 2> gpurun_out/${T}_sq1.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${T}_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra > gpurun_out/${T}_sq2.json 2> gpurun_out/${T}_sq2.err || exit 1
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(float)
for d in ("${T}_sq1", "${T}_sq2"):
    for fn in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(fn)):
            if "k_paths4" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({"kernel": "k_paths4", "launches": 1, "counters": dict(tot)}, open("gpurun_out/${T}_sq_k_paths4.json", "w"), indent=1)
PY
rm -rf gpurun_out/${T}_sq1 gpurun_out/${T}_sq2
echo sq done
rm -f gpurun_out/${T}_stats/*/*kernel_trace.csv
timeout -k 10 200 python3 bench.py --workload recsim --steps 3 --warmup 1 > gpurun_out/${T}_recsim.json 2> gpurun_out/${T}_recsim.err || exit 1
timeout -k 10 200 python3 bench.py --workload dense --steps 3 --warmup 1 > gpurun_out/${T}_dense.json 2> gpurun_out/${T}_dense.err || exit 1
timeout -k 10 200 python3 bench.py --workload c1 --steps 3 --warmup 1 > gpurun_out/${T}_c1.json 2> gpurun_out/${T}_c1.err || exit 1
echo extra done
