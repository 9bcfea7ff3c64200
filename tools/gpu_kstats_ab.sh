#!/bin/bash
# usage: gpu_kstats_ab.sh <outdir> <grep pattern> variant1 variant2 ...   (variant = "NAME:ENV1=V1,ENV2=V2")
# per variant: rocprofv3 kernel statistics of a short bench run, the rows matching the pattern; then a quiet bench line
O=$GRAFT_REPO_ROOT/gpurun_out/$1; PAT=$2; shift 2; mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  envcmd="env"; IFS=',' read -ra E <<< "$envs"; for e in "${E[@]}"; do [[ -n "$e" ]] && envcmd="$envcmd $e"; done
  cd /tmp
  $envcmd rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/kt_$name.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$name/*/*kernel_stats.csv | head -1) $O/kstats_$name.csv; rm -rf $O/kt_$name
  echo "== $name"; python tools/kstats_show.py $O/kstats_$name.csv $PAT
  $envcmd python bench.py --no-secondary --no-cpu-baseline > $O/bench_$name.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('   bench', d['value'], d['ms_per_step'])"
done
