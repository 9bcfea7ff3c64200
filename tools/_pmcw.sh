cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -i tools/pmc_hbm.txt --kernel-trace --output-format csv -d gpurun_out/pmcw -- python tools/train_step_bench.py --steps 1 > gpurun_out/pmcw.log 2>&1
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcw/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'wgrad_kernel<128, 128' in k or 'igemm_conv_kernel<3, 1, 128, 0, 1' in k:
            agg[(k[28:75],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
