R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4fin7_resnet; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
W=/tmp/rs; mkdir -p $W; cd $W
PYTHONPATH=$R/tests python -c "import golden.gen_resnet_csv as g; g.write_case('.',3,20,16)"
HCONV_IMAGE_BATCH=8 HCONV_SEED=11 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $R/optimal_conv_amd/host/conv --test-mode resnet 3 20 1 16 false > $O/run.log 2>&1
grep "Total done" $O/run.log; head -12 $O/stats/*/run_kernel_stats.csv 2>/dev/null | cut -c1-160 || find $O -name "*kernel_stats*"
