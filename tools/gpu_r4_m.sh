#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
for b in 1 47; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$b -- python $GRAFT_REPO_ROOT/tools/dsac_bench.py $b > $O/kt_$b.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$b/*/*kernel_stats.csv | head -1) $O/kstats_$b.csv; rm -rf $O/kt_$b
  echo "== B=$b"; python tools/kstats_show.py $O/kstats_$b.csv dsac; tail -1 $O/kt_$b.log
done
