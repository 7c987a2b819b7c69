#!/bin/bash
# usage: sq_a.sh TAG  (GPU box, repo root): SQ counters of the stage-A kernels (two passes of 8 counters; kernels run one at a
# time under --pmc, so these are per-kernel figures without the overlap of the class launches) -> gpurun_out/TAG_sq_stage_a.json
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/${T}_sq1 -- python3 profiles/tools/run_a.py 2 > gpurun_out/${T}_sq1.txt 2> gpurun_out/${T}_sq1.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${T}_sq2 -- python3 profiles/tools/run_a.py 2 > gpurun_out/${T}_sq2.txt 2> gpurun_out/${T}_sq2.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM --output-format csv -d gpurun_out/${T}_sq3 -- python3 profiles/tools/run_a.py 2 > gpurun_out/${T}_sq3.txt 2> gpurun_out/${T}_sq3.err || echo "third pass failed (counter names)"
python3 - <<PY
import csv, glob, collections, json, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for d in ("${T}_sq1", "${T}_sq2", "${T}_sq3"):
    for fn in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(fn)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void xmap::", "").replace("xmap::", "")
            if not k.startswith("k_"): continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]) / 2     # per pass (two passes run)
out = {k: dict(v) for k, v in tot.items() if v.get("SQ_BUSY_CYCLES", 0) > 0}
json.dump(out, open("gpurun_out/${T}_sq_stage_a.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    w = max(v.get("SQ_WAVES", 1), 1)
    print("%-46s waves %8d  per wave: VALU %6.0f SALU %6.0f LDS %5.0f VMEMR %5.1f VMEMW %5.1f  wave_cycles %7.0f  active VALU cyc %6.0f LDS cyc %6.0f  lanes/VALU %.1f" % (
        k[:46], w, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w,
        v.get("SQ_INSTS_VMEM_WR", 0) / w, v.get("SQ_WAVE_CYCLES", 0) / w, v.get("SQ_ACTIVE_INST_VALU", 0) / w, v.get("SQ_ACTIVE_INST_LDS", 0) / w,
        v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_ACTIVE_INST_VALU", 1), 1)))
PY
rm -rf gpurun_out/${T}_sq1 gpurun_out/${T}_sq2 gpurun_out/${T}_sq3
