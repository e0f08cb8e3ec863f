#!/bin/bash
# Collects everything profiles/ holds for one round, on the GPU box, from the repo root:
#   bash profiles/collect.sh <tag>        -> gpurun_out/profiles_<tag>/  (copy what is to be judged into profiles/)
# 1. the bench line (un-profiled)                                        <tag>_bench_line.json
# 2. rocprofv3 --kernel-trace --stats of the same command                 <tag>_kernel_stats_bench_n1.csv
#    + the per-launch trace of gicp_align_kernel (every dispatch)         <tag>_per_launch_trace.csv / .txt
# 3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (collect_traffic)   <tag>_pmc_*_summary.csv, traffic.json
# rocprofv3 wraps `python3 bench.py` directly (no env/bash hop); counters never share a run with --stats.
set -e
TAG=${1:-rXX}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
echo "bench done"
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $ROOT/bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-example > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/rocprof.err
cp $(grep -l gicp_align_kernel $(find /tmp/prof_$TAG -name "*kernel_stats.csv") | head -1) $OUT/${TAG}_kernel_stats_bench_n1.csv
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
f = max(glob.glob(f"/tmp/prof_{tag}/**/*kernel_trace.csv", recursive=True), key=lambda p: sum("gicp_align_kernel" in l for l in open(p)))
rows = [r for r in csv.DictReader(open(f)) if "gicp_align_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[20:20 + 31 * 20]  # skip the 20 warm-up launches; keep the 31 timed blocks (one 20-launch alignment each)
with open(f"{out}/{tag}_per_launch_trace.csv", "w") as o:
    o.write("dispatch,launch_in_alignment,start_ns,duration_us,gap_to_previous_us,vgpr,workgroup,grid\n")
    prev_end = None
    for i, r in enumerate(rows):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:.2f}"
        o.write(f"{i},{i % 20},{st},{(en - st) / 1e3:.2f},{gap},{r.get('VGPR_Count', '')},{r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))},{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\n")
        prev_end = en
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n_al = len(d) // 20
with open(f"{out}/{tag}_per_launch_trace.txt", "w") as o:
    o.write(f"gicp_align_kernel: {len(d)} dispatches = {n_al} alignments of 20 launches (rocprofv3 --kernel-trace)\n")
    o.write("mean duration by launch index within an alignment (us):\n")
    for k in range(20):
        v = [d[a * 20 + k] for a in range(n_al)]
        o.write(f"  launch {k:2d}: mean {sum(v) / len(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f}\n")
    o.write(f"all launches: mean {sum(d) / len(d):.2f} us\n")
print(open(f"{out}/{tag}_per_launch_trace.txt").read())
PY
python3 $ROOT/profiles/collect_traffic.py > $OUT/collect_traffic.log 2>&1 || echo "collect_traffic failed"
cp $ROOT/gpurun_out/traffic.json $OUT/traffic.json 2>/dev/null || true
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(ls -t $ROOT/gpurun_out/pmc_traffic/bench/$c/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" "$c" > $OUT/${TAG}_pmc_${c}_summary.csv <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"][:100]].append(float(r["Counter_Value"]))
print("kernel,launches,mean_KiB,min_KiB,max_KiB")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"\"{k}\",{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f}")
PY
done
echo "collected into $OUT"
