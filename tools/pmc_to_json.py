"""Turns a tools/prof_pmc.sh summary (gpurun_out/<tag>_summary.txt) into profiles/pmc_latest.json, the PMC-derived
constants bench.py prints beside its live timing (roofline.traffic and roofline.decomposition).
usage: python tools/pmc_to_json.py gpurun_out/r02_pmc_summary.txt [kernel-substring] [kernel_ms of the profiled launches] [output name under profiles/]"""
import subprocess
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "poolkernel"
sec, cur = {}, None
for line in open(src):
    if line.startswith("== "):
        cur = line[3:].strip()
        sec[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+) mean=(\S+)", line)
        if m and cur:
            sec[cur][m.group(1)] = float(m.group(3))
name = next(k for k in sec if want in k and "<true>" not in k.replace("finalize<true>", ""))   # (not the instrumented build)
c = sec[name]
out = {
    "_source": f"{os.path.relpath(src, ROOT)}: rocprofv3 --pmc passes (tools/prof_pmc.sh; FETCH_SIZE and WRITE_SIZE in separate passes) of "
               "`bench.py --steps 1 --warmup 1`, mean over the launches of the kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 tallies 128-B read requests at 64 B); units KiB -> bytes; Infinity-Cache hits are counted, so this is an "
               "upper bound on HBM bytes",
    "kernel": name,
    "valu_wave_insts_per_launch": c["SQ_INSTS_VALU"],
    "valu_lane_utilisation": c["VALUUtilization"] / 100.0 if "VALUUtilization" in c else c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0),
    "salu_insts_per_launch": c.get("SQ_INSTS_SALU"), "lds_insts_per_launch": c.get("SQ_INSTS_LDS"),
    "mean_occupancy_waves_per_cu": c.get("MeanOccupancyPerCU"),
    "grbm_gui_active": c.get("GRBM_GUI_ACTIVE"),
    "FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": int(c["WRITE_SIZE"] * 1024 + 2 * c["FETCH_SIZE"] * 1024) if "WRITE_SIZE" in c and "FETCH_SIZE" in c else None,
    "algorithmic_bytes_per_launch": 1024 * 768 * 12,
    # provenance for bench.py's roofline.from_committed_profile: the kernel time the counters were collected at and the source they belong to
    "kernel_ms": float(sys.argv[3]) if len(sys.argv) > 3 else (c["GRBM_GUI_ACTIVE"] / (8 * 2.4e6) if "GRBM_GUI_ACTIVE" in c else None),
    "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
}
json.dump(out, open(os.path.join(ROOT, "profiles", sys.argv[4] if len(sys.argv) > 4 else "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
