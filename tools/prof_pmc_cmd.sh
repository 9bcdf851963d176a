#!/bin/bash
# PMC counters for an arbitrary short command (separate passes; kernel-trace only, as the pool requires).
# usage (on the GPU box, from the repo root): bash tools/prof_pmc_cmd.sh <tag> python tools/bench_grid.py 64 0x0 --nocheck
# Passes: SPT_PMC_ONLY="0 1" restricts them.  Summary -> gpurun_out/<tag>_summary.txt (mean per kernel over its launches).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"
 "GRBM_GUI_ACTIVE WRITE_SIZE"
 "FETCH_SIZE"
 "VALUBusy VALUUtilization"
 "MeanOccupancyPerCU OccupancyPercent"
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH"
)
i=0
for P in "${PASSES[@]}"; do
  if [ -n "${SPT_PMC_ONLY:-}" ] && ! echo " $SPT_PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  ( cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${TAG}_p$i -- "$@" > $R/gpurun_out/${TAG}_p$i.log 2>&1 ) || { echo "pass $i failed"; tail -5 $R/gpurun_out/${TAG}_p$i.log; }
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
        if "rocclr" in k: continue
        out.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:28s} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open(f"{R}/gpurun_out/{TAG}_summary.txt").read())
PY
