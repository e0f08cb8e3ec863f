#!/bin/bash
# Everything profiles/ holds for round 5, on the GPU box from the repo root: bash scratch/collect_r05.sh <tag>
TAG=${1:-r05_k}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
bash $ROOT/profiles/collect.sh $TAG > $OUT/collect.log 2>&1
echo "collect.sh done"; tail -3 $OUT/collect.log
for w in 2 4 8; do
  python3 $ROOT/bench.py --emulate-world $w --no-cpu-baseline --no-stages > $OUT/${TAG}_bench_line_emulated_rank_of_$w.json 2> $OUT/emu$w.err
  echo "emulated rank of $w done"
done
for x in direct rccl-row torch-row; do
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python3 $ROOT/bench.py --force-sharded --exchange $x --no-cpu-baseline --no-stages > $OUT/${TAG}_bench_line_one_rank_$x.json 2> $OUT/onerank_$x.err
  echo "one rank $x done"
done
python3 $ROOT/bench.py --reg p2d --no-cpu-baseline --no-stages > $OUT/${TAG}_bench_line_p2d.json 2> $OUT/p2d.err
echo "p2d done"
$ROOT/tests/cpp/example_registration $ROOT/tests/golden/source.ply $ROOT/tests/golden/target.ply 100 10 > $OUT/${TAG}_example_registration.txt 2>&1
echo "example done"
python3 - $OUT $TAG <<'PY'
import json, sys, glob, os
out, tag = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"{out}/{tag}_bench_line*.json")):
    try:
        d = json.loads([x for x in open(f) if x.startswith("{")][-1])
        L = d.get("launches_of_one_alignment") or []
        print(os.path.basename(f), "step %.2f us" % (1e3 * d["ms_per_step"]), "value %.3e" % d["value"],
              "launches", [x["us"] for x in L][:6], "until-conv", (d.get("until_converged") or {}).get("ms_per_alignment"),
              "roofline frac", round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
