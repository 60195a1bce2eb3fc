#!/bin/bash
# bench.py's extras watchdog: (1) the default run, (2) two gloo ranks sharing the one GPU (the N > 1 path's dry run), (3) the same with a 3 s watchdog: the headline must still be printed
cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-r4wd}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python bench.py --steps 10 --warmup 3 > $O/n1.json 2> $O/n1.err; echo "n1 exit $?"
HC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $O/n2.json 2> $O/n2.err; echo "n2 exit $?"
HC_BENCH_EXTRAS_TIMEOUT=3 HC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 > $O/n2_wd.json 2> $O/n2_wd.err; echo "n2 watchdog exit $?"
for f in n1 n2 n2_wd; do python - <<PY
import json
try:
    lines=[l for l in open("$O/$f.json") if l.startswith("{")]
    d=json.loads(lines[-1]); print("$f", len(lines), "line(s):", round(d["value"],1), d["n_gpus"], sorted(k for k in d if k in ("workloads","sharded_conv","extras","cpu_baseline")), d.get("extras"))
except Exception as e: print("$f", "FAILED", e); print(open("$O/$f.err").read()[-1500:])
PY
done
