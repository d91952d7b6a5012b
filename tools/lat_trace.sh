export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lat -- python $R/tools/lat_profile.py ) > gpurun_out/lat_prof.log 2>&1
grep staged gpurun_out/lat_prof.log
find /tmp/prof_lat -name "*kernel_trace.csv" -exec cp {} /tmp/lat_trace.csv \;
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('/tmp/lat_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'fbank' in r['Kernel_Name']]
seg=rows[idx[-1]:]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
span=int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp'])
print('last run: kernels',len(seg),'busy us',busy/1e3,'span us',span/1e3)
gaps=[int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']) for i in range(len(seg)-1)]
import statistics
print('gap median ns', statistics.median(gaps), 'mean', sum(gaps)/len(gaps), 'max', max(gaps))
c=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name'][:46]; c[n][0]+=1; c[n][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,(k,t) in sorted(c.items(),key=lambda x:-x[1][1])[:12]: print('%-48s %4d %8.1f us  avg %.2f'%(n,k,t/1e3,t/1e3/k))
PY
