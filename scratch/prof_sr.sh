cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_sr
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sr -o run -- python $R/scratch/sr_bench.py $1 > $R/gpurun_out/prof_sr.log 2>&1
f=$(find $R/gpurun_out/prof_sr -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
python - <<PY
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_sr/**/*kernel_trace.csv',recursive=True)[0]
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k4_conv' in r['Kernel_Name'] or 'k4_sft' in r['Kernel_Name']:
        d[(r['Kernel_Name'].split('(')[0], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    print(k, len(v), 'total_us', round(sum(v)), 'avg_us', round(sum(v)/len(v),1))
PY
rm -rf $R/gpurun_out/prof_sr
