export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ramp
BENCH_CHILD_STEPS=30 BENCH_CHILD_PIPELINED=1 rocprofv3 --kernel-trace -d /tmp/ramp -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ramp/**/*kernel_trace.csv', recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ','').split('<')[0]
    per[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k in ('lowpass_down_kernel', 'dog_scan_all_kernel', 'descr_all_kernel', 'orient_all_gather_kernel'):
    v = sorted(per[k])
    d = [(b - a) / 1e6 for a, b in v]
    if k == 'dog_scan_all_kernel':
        d = [d[i] + d[i + 1] for i in range(0, len(d) - 1, 2)]
    print(k, ' '.join('%.3f' % x for x in d))
v = sorted(per['lowpass_down_kernel'])
print('step period', ' '.join('%.3f' % ((v[i + 1][0] - v[i][0]) / 1e6) for i in range(len(v) - 1)))
PY
