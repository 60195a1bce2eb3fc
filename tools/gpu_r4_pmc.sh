#!/bin/bash
# round 4: (1) counters of `conv 3 3` (one context, the benched launch-set size) -> traffic per conv, per-kernel table, VALU busy / class-priced issue floor;
# (2) fabric traffic and kernel stats of the convReLU 5 1 tail at HCONV_IMAGE_BATCH=$NBCHAIN. Separate rocprofv3 passes, --kernel-trace + --pmc only.
# usage: OUT=name bash tools/gpu_r4_pmc.sh ; results under gpurun_out/$OUT/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4pmc}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "VALU|SQ_BUSY|SQ_INST_CYCLES" | head -60 > $O/counters_available.txt)
if [ "${SKIP_CONV:-0}" != "1" ]; then
S=1; NB=${BATCH:-4}; K=2; W=1
BENCH="python $R/bench.py --steps $K --warmup $W --batch $NB --streams $S --no-cpu-baseline --no-workloads"
CONVS=$(( (1 + W + K) * S * NB + (K < 3 ? K : 3) * NB + 10 ))
echo "$BENCH ; convolutions in the run: $CONVS" > $O/command.txt
run_pmc() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o run -- $BENCH > $O/$name.log 2>&1); }
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $BENCH > $O/stats.log 2>&1)
(cd $R && PMC_COMMAND="bench.py --steps $K --warmup $W --batch $NB --streams $S" PMC_CONVS=$CONVS python tools/pmc_table.py gpurun_out/${OUT:-r4pmc} > $O/table.txt 2>&1; cat $O/table.txt)
python $R/tools/pmc_traffic.py $O/fetch/run_counter_collection.csv $O/write/run_counter_collection.csv $CONVS $O/traffic.json | tee $O/traffic.txt
python $R/tools/valu_floor.py combine $O $CONVS $O/valu.json | tee $O/valu.txt
fi
# (2) the chain
NBC=${NBCHAIN:-4}; IT=2
[ "${SKIP_CHAIN:-0}" = "1" ] && exit 0
W2=/tmp/r4pmc_chain; mkdir -p $W2; cd $W2
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
bench._write_conv_csv("test_conv_data", 5, 1, $IT)
PY
CH="$R/optimal_conv_amd/host/conv convReLU 5 1 $IT"
for pmc in FETCH_SIZE WRITE_SIZE; do
  HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 900 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/chain_$pmc -o run -- $CH > $O/chain_$pmc.log 2>&1
done
HCONV_IMAGE_BATCH=$NBC HCONV_SKIP_BL=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain_stats -o run -- $CH > $O/chain_stats.log 2>&1
python - <<PY
import csv, collections, json, glob
O = "$O"; layers = $IT; nb = $NBC
chain = lambda k: any(t in k for t in ("_mm", "hc_k_lv_", "ks_", "permute", "mod_raise", "qp_", "basis_yv", "lincomb", "rotate_finish"))
def tot(pmc, corr):
    acc = collections.defaultdict(float); n = collections.Counter()
    for path in glob.glob(O + "/chain_%s/**/run_counter_collection.csv" % pmc, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == pmc:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if chain(k): acc[k] += float(r["Counter_Value"]) * 1024 * corr; n[k] += 1
    return acc, n
rd, n = tot("FETCH_SIZE", 2.0); wr, _ = tot("WRITE_SIZE", 1.0)
per = (sum(rd.values()) + sum(wr.values())) / layers / nb
res = {"bytes_per_ciphertext_layer": per, "bytes_per_layer_launch_set": per * nb, "read_bytes_per_ciphertext_layer": sum(rd.values()) / layers / nb, "write_bytes_per_ciphertext_layer": sum(wr.values()) / layers / nb,
       "per_kernel_per_ciphertext_layer": {k: {"read": rd.get(k, 0) / layers / nb, "write": wr.get(k, 0) / layers / nb, "launches_per_layer": n.get(k, 0) / layers} for k in sorted(set(rd) | set(wr))},
       "measured_with": {"command": "HCONV_IMAGE_BATCH=%d HCONV_SKIP_BL=1 conv convReLU 5 1 %d" % (nb, layers), "ciphertexts_per_launch_set": nb, "layers_in_run": layers,
                         "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950); KiB units; the chain's kernels only (key generation, DFT-matrix encoding and the convolution excluded)"}}
json.dump(res, open(O + "/traffic_convrelu_5_1_n%d.json" % nb, "w"), indent=1)
print("chain traffic per ciphertext-layer at n = %d: %.1f GB (read %.1f, write %.1f)" % (nb, per / 1e9, res["read_bytes_per_ciphertext_layer"] / 1e9, res["write_bytes_per_ciphertext_layer"] / 1e9))
PY
ls $O
