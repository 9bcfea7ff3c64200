# PMC passes over the stand-alone 1x1 split conv (tools/conv_bench.py --split): kernel trace only, separate --pmc passes
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/split1x1pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_split.txt --kernel-trace --output-format csv -d $OUT/f -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --k 1 --B 44 --stats --norm --split --iters 5 > $OUT/f.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/f $OUT/f.csv
rm -rf $OUT/f
grep -h "split_conv1x1" $OUT/f.csv | cut -d, -f1,5,6,7,10
