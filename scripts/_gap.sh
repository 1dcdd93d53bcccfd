cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/sess_kt -o s -- python /root/repo/scripts/run_session_config1.py --oracle-views 0 --out /root/repo/gpurun_out/session_traced.json > /tmp/sess.log 2>&1
tail -2 /tmp/sess.log | cut -c1-300
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/sess_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.split('(')[0].replace('void ', '')[:60]
gaps = collections.Counter(); cnt = collections.Counter()
busy = 0; t_end = None; first = int(rows[0]['Start_Timestamp']); last = 0
# restrict to the mapping phase: after the first 20% of time? report all, plus per pair
for a, b in zip(rows, rows[1:]):
    ea, sb = int(a['End_Timestamp']), int(b['Start_Timestamp'])
    busy += int(a['End_Timestamp']) - int(a['Start_Timestamp'])
    g = sb - ea
    if g > 15000:
        k = (short(a['Kernel_Name']), short(b['Kernel_Name']))
        gaps[k] += g; cnt[k] += 1
total = int(rows[-1]['End_Timestamp']) - first
print('kernels', len(rows), 'span_s', total / 1e9, 'busy_s', busy / 1e9, 'gaps>15us_s', sum(gaps.values()) / 1e9)
for k, v in gaps.most_common(25):
    print('%8.1f ms  n=%5d  avg %7.1f us   %s  ->  %s' % (v / 1e6, cnt[k], v / cnt[k] / 1e3, k[0], k[1]))
PY
