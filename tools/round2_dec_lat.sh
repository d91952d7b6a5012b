export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_baseline_sizes.py tests/test_gpu_seaco.py tests/test_gpu_online.py tests/test_gpu_timestamp.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5
for v in 1 0; do
  echo "PF_DEC_H32=$v"
  PF_DEC_H32=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb=d['kernel_breakdown_ms_per_step']
print(round(d['ms_per_step'],3), d['ids_sha1'][:8], {k:round(v['ms'],3) for k,v in kb.items() if ('dec' in k or k in ('layernorm','fsmn','attn_cross')) })
"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-200
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lat -- python $R/tools/lat_profile.py ) > gpurun_out/lat_prof.log 2>&1
grep staged gpurun_out/lat_prof.log
find /tmp/prof_lat -name "*kernel_stats.csv" -exec cp {} gpurun_out/lat_kernel_stats.csv \;
find /tmp/prof_lat -name "*kernel_trace.csv" -exec cp {} /tmp/lat_trace.csv \;
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/lat_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last run = kernels after the last fbank launch
idx=[i for i,r in enumerate(rows) if 'fbank' in r['Kernel_Name']]
seg=rows[idx[-1]:]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
span=int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp'])
print('last run: kernels',len(seg),'busy us',busy/1e3,'span us',span/1e3)
gaps=[(int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']),seg[i]['Kernel_Name'][:40],seg[i+1]['Kernel_Name'][:40]) for i in range(len(seg)-1)]
gaps.sort(reverse=True)
print('largest gaps (ns):',gaps[:6])
import collections
c=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name'][:50]; c[n][0]+=1; c[n][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,(k,t) in sorted(c.items(),key=lambda x:-x[1][1])[:14]: print('%-52s %4d %8.1f us  avg %.1f'%(n,k,t/1e3,t/1e3/k))
PY
