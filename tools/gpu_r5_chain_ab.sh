#!/bin/bash
# round 5: A/B of the convReLU 5 1 tail through the product CLI. CONFIGS = space-separated "name[:ENV=VAL[,ENV=VAL...]][@libdir]" (libdir under tools/_variants/ holding a libhconv.so;
# none = the product's); NB = HCONV_IMAGE_BATCH values; REPS interleaved repetitions. Prints ms per launch set of the steady layers (the first allocates the pools).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r5cab}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
W=/tmp/r5cab; mkdir -p $W; cd $W
IT=${ITERS:-4}
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',5,1,i) for i in range($IT)]"
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in ${CONFIGS:-base}; do
    name=${cfg%%[:@]*}; envs=""; lib=""
    case "$cfg" in *@*) lib=${cfg##*@};; esac
    case "$cfg" in *:*) envs=${cfg#*:}; envs=${envs%%@*};; esac
    for nb in ${NB:-4}; do
      ( [ -n "$lib" ] && export LD_LIBRARY_PATH=$R/tools/_variants/$lib:${LD_LIBRARY_PATH:-}
        for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done
        HCONV_IMAGE_BATCH=$nb HCONV_SKIP_BL=1 HCONV_SEED=7 HCONV_PROFILE=${PROFILE:-0} timeout 900 $R/optimal_conv_amd/host/conv --test-mode convReLU 5 1 $IT > $O/${name}_nb${nb}_r$rep.log 2>&1 )
      python3 - "$name rep $rep n=$nb" $O/${name}_nb${nb}_r$rep.log $nb <<'PY'
import re, sys
t = open(sys.argv[2]).read(); nb = int(sys.argv[3])
def secs(tok):
    m = re.match(r"([0-9.e+-]+)(µs|ms|s)$", tok); return float(m.group(1)) * {"µs": 1e-6, "ms": 1e-3, "s": 1.0}[m.group(2)]
conv = [secs(x) for x in re.findall(r"^Conv \(with BN\) Done in (\S+) ", t, re.M)]
ctos = [secs(x) for x in re.findall(r"^Done in (\S+) $", t, re.M)][-len(conv):] if conv else []
relu = [secs(x) for x in re.findall(r"ReLU Done in (\S+) ", t, re.M)]
stoc = [secs(x) for x in re.findall(r"^Boot \(StoC\) Done in (\S+) ", t, re.M)]
prec = re.findall(r"MED Prec : \((\S+),", t)
n = min(len(conv), len(ctos), len(relu), len(stoc))
if n < 2: print(sys.argv[1], "FAILED", t[-300:].replace("\n", " | ")); sys.exit(0)
L = [1e3 * (conv[i] + ctos[i] + relu[i] + stoc[i]) for i in range(1, n)]
print("%-28s layer %7.2f ms (min %7.2f) = %6.2f ms per ciphertext-layer | ctos+sine %6.2f relu %6.2f stoc %5.2f | MED prec %s" % (sys.argv[1], sum(L) / len(L), min(L), sum(L) / len(L) / nb,
      1e3 * sum(ctos[1:n]) / (n - 1), 1e3 * sum(relu[1:n]) / (n - 1), 1e3 * sum(stoc[1:n]) / (n - 1), prec[-1] if prec else "?"))
PY
    done
  done
done
