#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 4 --warmup 1"
for t in 1 2 3 5 15; do
  cd /tmp
  XL_WINO_OUT_TPB=$t rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$t -- $B > $O/kt_$t.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$t/*/*kernel_stats.csv | head -1) $O/kstats_$t.csv; rm -rf $O/kt_$t
  echo "== tpb $t"; python tools/kstats_show.py $O/kstats_$t.csv wino6_out_kernel gn_final
done
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
