#!/bin/bash
# HBM-side traffic of the convReLU chain kernels (the *_mm / lv_* / ks_* / rescale_* kernels of one `convReLU 5 1 2` run):
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only) -> gpurun_out/relu_traffic/
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/relu_traffic; mkdir -p $O
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=/tmp/relutraffic; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',5,1,i) for i in range(2)]"
for pmc in FETCH_SIZE WRITE_SIZE; do
  HCONV_SKIP_BL=1 HCONV_SEED=7 timeout 900 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$pmc -o run -- $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 2 > $O/run_$pmc.log 2>&1
done
python - <<PY
import csv, collections, json, glob
def tot(pmc, corr):
    acc = collections.defaultdict(float); n = collections.Counter()
    for path in glob.glob("$O/pmc_%s/**/run_counter_collection.csv" % pmc, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == pmc:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "_mm" in k or "hc_k_lv_" in k or "ks_" in k or "permute" in k or "mod_raise" in k or "rows_fwd_mac" in k:
                    acc[k] += float(r["Counter_Value"]) * 1024 * corr; n[k] += 1
    return acc, n
rd, n = tot("FETCH_SIZE", 2.0); wr, _ = tot("WRITE_SIZE", 1.0)
layers = 2
res = {"layers_in_run": layers, "bytes_per_layer": (sum(rd.values()) + sum(wr.values())) / layers, "read_bytes_per_layer": sum(rd.values()) / layers, "write_bytes_per_layer": sum(wr.values()) / layers,
       "per_kernel": {k: {"read": rd.get(k, 0) / layers, "write": wr.get(k, 0) / layers, "launches": n.get(k, 0) / layers} for k in sorted(set(rd) | set(wr))},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950); KiB units; chain kernels only (key generation's per-limb kernels excluded)"}
json.dump(res, open("$O/traffic_convrelu_5_1.json", "w"), indent=1)
print("chain traffic per convReLU layer: %.1f GB (read %.1f, write %.1f)" % (res["bytes_per_layer"] / 1e9, res["read_bytes_per_layer"] / 1e9, res["write_bytes_per_layer"] / 1e9))
PY
