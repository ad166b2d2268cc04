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
rounds = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")) for r in rows if r["Kernel_Name"].startswith("icv::k_ward_round")]
rounds.sort()
with open("gpurun_out/ward_prof/rounds.txt", "w") as o:
    for i, (_, us, g) in enumerate(rounds):
        o.write(f"round {i:3d} grid {g:>10s} {us:10.1f} us\n")
    o.write(f"total {sum(u for _, u, _ in rounds) / 1e3:.1f} ms over {len(rounds)} rounds\n")
print(open("gpurun_out/ward_prof/rounds.txt").read()[-1500:])
s = glob.glob("gpurun_out/ward_prof/**/ward_kernel_stats.csv", recursive=True)[0]
print(open(s).read()[:1500])
PY
