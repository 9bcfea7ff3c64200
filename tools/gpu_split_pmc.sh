cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/splitpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_split.txt --kernel-trace --output-format csv -d $OUT/f -- python $GRAFT_REPO_ROOT/tools/split_gemm_bench.py 44 > $OUT/f.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/f $OUT/f.csv
rm -rf $OUT/f
grep -h "split_gemm_persist" $OUT/f.csv | cut -d, -f1,2,5,6,7,10
