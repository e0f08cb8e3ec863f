# bench.py's N > 1 branches on the one-GPU box: two ranks share cuda:0, gloo carries torch.distributed (SP_BENCH_SHARE_GPU=1;
# the row goes through torch.distributed, eagerly). A rehearsal of the code path, not a measurement.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rehearse
for v in "--warmup 5 --steps 20" "--warmup 20 --steps 100 --repeats 5"; do
SP_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 $v 2> gpurun_out/rehearse/err.log | tee gpurun_out/rehearse/line.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v', 'n_gpus', d['n_gpus'], 'pose_err', d['pose_max_abs_err_vs_ground_truth'], 'inliers', d['inliers_last_iteration'], 'ms/step', round(d['ms_per_step'],4), [(l['carrier'], round(l['ms_per_step'],4)) for l in d['exchange_legs']], d['config']['exchange_verification_legs'])" || exit 1
done
