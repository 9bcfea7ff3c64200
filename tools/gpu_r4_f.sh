#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2"
for v in t4 t8 old; do
  cd /tmp
  unset XL_STEM12_TILE XL_NO_STEM12
  [ $v = t8 ] && export XL_STEM12_TILE=8
  [ $v = old ] && export XL_NO_STEM12=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- $B > $O/kt_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$v/*/*kernel_stats.csv | head -1) $O/kstats_$v.csv
  cp $(ls $O/kt_$v/*/*kernel_trace.csv | head -1) $O/ktrace_$v.csv
  rm -rf $O/kt_$v
  echo "== $v"; python tools/kstats_show.py $O/kstats_$v.csv stem12 conv1_mfma s2_kernel gn_stats dsac
  tail -1 $O/kt_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['cnn_ms_per_batch'])"
done
