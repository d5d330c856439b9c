cd $GRAFT_REPO_ROOT
T0=$(date +%s)
python bench.py > gpurun_out/bench_sub.json 2> gpurun_out/bench_sub.err
echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python -c "
import json;d=json.load(open('gpurun_out/bench_sub.json'));print(d['value'],d['ms_per_step']);print(json.dumps(d['roofline']['train_step'],indent=0)[:1500])"
