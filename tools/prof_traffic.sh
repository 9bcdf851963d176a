#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of tools/traffic_probe.py for the two scene variants.  usage: bash tools/prof_traffic.sh <tag>
set -u
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for S in cornell9 noglass; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_${S}_$C -- python $R/tools/traffic_probe.py $S > $R/gpurun_out/${TAG}_${S}_$C.log 2>&1 || echo "pass $S $C failed"
  done
done
python3 - "$R" "$TAG" <<'PY'
import csv, glob, sys
R, TAG = sys.argv[1], sys.argv[2]
for S in ("cornell9", "noglass"):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{R}/gpurun_out/{TAG}_{S}_{C}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "poolkernel" in row["Kernel_Name"] or "finalize" in row["Kernel_Name"]:
                    v = float(row["Counter_Value"])
                    gb = v * 1024 * (2 if C == "FETCH_SIZE" else 1) / 1e9
                    print(f"{S:9s} {row['Kernel_Name'].split('(')[0][:40]:40s} {C:10s} {v:14.0f} KiB -> {gb:8.3f} GB" + (" (doubled)" if C == "FETCH_SIZE" else ""))
PY
