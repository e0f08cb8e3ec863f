cd $GRAFT_REPO_ROOT
for p in 0.5 0.35 0.7 1.0 1.5 0.5; do
python bench.py --repeats 9 --no-cpu-baseline --no-stages --ppc $p 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); l=d['launches_of_one_alignment']
print('ppc $p', [x['us'] for x in l[:5]], 'us/step', round(d['ms_per_step']*1e3,2), 'conv', round(d['until_converged']['ms_per_alignment'],4), 'searched', [x['searched_points'] for x in l[:4]])"
done
