#!/bin/bash
# rocprofv3 kernel trace of the config-5 path (run on the GPU box): per-kernel stats + per-round durations.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-100000}
mkdir -p $REPO/gpurun_out/ward_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/ward_prof -o ward -- \
    python $REPO/tools/bench_ward.py --cells $N > $REPO/gpurun_out/ward_prof/bench.log 2>&1 || true
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/ward_prof/**/ward_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def sel(prefix):
    v = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) // 256)
         for r in rows if prefix in r["Kernel_Name"]]
    v.sort()
    return v
scan, merge, pairs = sel("k_ward_scan"), sel("k_ward_merge"), sel("k_ward_pairs")
push, compact = sel("k_ward_push"), sel("k_ward_compact")
with open("gpurun_out/ward_prof/rounds.txt", "w") as o:
    o.write("round: rows searched, us | rows merged, us | pairs kernel us\n")
    for i, (_, us, g) in enumerate(scan):
        m = merge[i - 1] if 0 < i <= len(merge) else (0, 0.0, 0)
        pk = pairs[i][1] if i < len(pairs) else 0.0
        o.write(f"round {i:3d} scan {g:7d} {us:9.1f} | merge {m[2]:6d} {m[1]:9.1f} | pairs {pk:7.1f}\n")
    o.write(f"total scan {sum(u for _, u, _ in scan) / 1e3:.1f} ms, merge {sum(u for _, u, _ in merge) / 1e3:.1f} ms, "
            f"pairs {sum(u for _, u, _ in pairs) / 1e3:.1f} ms, strip push {sum(u for _, u, _ in push) / 1e3:.1f} ms "
            f"({len(push)} launches), compaction {sum(u for _, u, _ in compact) / 1e3:.1f} ms ({len(compact)} launches) "
            f"over {len(scan)} rounds\n")
print(open("gpurun_out/ward_prof/rounds.txt").read()[:6000])
s = glob.glob("gpurun_out/ward_prof/**/ward_kernel_stats.csv", recursive=True)[0]
print(open(s).read()[:1500])
PY
