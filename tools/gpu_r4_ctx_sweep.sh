cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4sw
for cfg in "4 4" "4 6" "4 8" "2 8" "3 6" "6 4" "2 12" "4 4"; do set -- $cfg
  python bench.py --steps 20 --warmup 5 --batch $1 --streams $2 --no-workloads --no-cpu-baseline > gpurun_out/r4sw/b$1_s$2.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r4sw/b$1_s$2.json'));print('batch $1 streams $2:', round(d['value'],1), 'conv/s', round(d['ms_per_step'],2),'ms/step')"
done
