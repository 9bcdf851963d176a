#!/bin/bash
# Instruction-mix PMC passes for the megakernel. usage: bash tools/prof_pmc2.sh <tag> [bench args]
set -u
TAG=${1:-pmcmix}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
 "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES"
 "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/${TAG}_p$i.log; }
  i=$((i+1))
done
python3 - "$R" "$TAG" <<'PY'
import csv, glob, sys, collections
R, TAG = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/{TAG}_p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{R}/gpurun_out/{TAG}_summary.txt", "w") as out:
    for k, d in acc.items():
        if "rocclr" in k or "finalize" in k: continue
        out.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:28s} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open(f"{R}/gpurun_out/{TAG}_summary.txt").read())
PY
