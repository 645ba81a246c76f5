# Dev (GPU box): per-chain step periods of the cascade's pipelined forward / backward from a per-dispatch kernel trace
# (start-to-start times of one marker kernel per chain, last iteration).
O=gpurun_out/r05/casc_trace; mkdir -p $O
export TMPDIR=/tmp
export AHEAD_GEMMS=14
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 scripts/dev_cascade_profile.py > $O/out.txt 2> $O/err.txt || echo "trace failed"
python3 - <<'P'
import csv, glob
f = glob.glob('gpurun_out/r05/casc_trace/t/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def starts(sub, n):
    xs = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if sub in r['Kernel_Name']]
    return xs[-n:]
import statistics
for name, sub in (('bottom fwd step (EpiGruZR 64x64)', 'igemm_kernel<unsigned short, 64, 64, 2, 2, 1, 1, rgp::EpiGruZR'),
                  ('top fwd step (EpiGruZR 128x32)', 'igemm_kernel<unsigned short, 128, 32, 4, 1, 4, 1, rgp::EpiGruZR'),
                  ('top bwd step (top_bwd1)', 'top_bwd1_kernel'), ('bottom bwd step (gru_bwd1)', 'gru_bwd1_kernel'),
                  ('feed-back step (EpiAtomicAdd ksplit 4, first of 3 per bottom step...)', 'top_bwd2_kernel')):
    xs = starts(sub, 35)
    d = [(b[0] - a[0]) / 1e3 for a, b in zip(xs, xs[1:])]
    print('%-60s span %.1f us, period median %.1f us (min %.1f max %.1f)' % (name, (xs[-1][1] - xs[0][0]) / 1e3, statistics.median(d), min(d), max(d)))
# the last backward: from the last-but-one maxout_bwd pair to the end
mb = [i for i, r in enumerate(rows) if 'maxout_bwd_kernel' in r['Kernel_Name']]
i0 = mb[-2]
t0 = int(rows[i0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows[i0:])
print('last backward: %.1f us, %d dispatches' % ((t1 - t0) / 1e3, len(rows) - i0))
# milestones inside it
def first_after(sub): 
    for r in rows[i0:]:
        if sub in r['Kernel_Name']: return (int(r['Start_Timestamp']) - t0) / 1e3
def last_end(sub):
    e = [int(r['End_Timestamp']) for r in rows[i0:] if sub in r['Kernel_Name']]
    return (max(e) - t0) / 1e3 if e else None
for sub in ('top_bwd1_kernel', 'gru_bwd1_kernel', 'gru_bwd2_kernel', 'pad_rows_kernel', 'wgrad_kernel', 'dense_colsum', 'rows_to'):
    print('  %-20s first start %s us, last end %s us' % (sub, first_after(sub), last_end(sub)))
# the calling stream's own dispatches of the last backward, and chain B's span
print('  calling stream (stream 0) in the last backward:')
prev = None
for r in rows[i0:]:
    if r.get('Stream_Id') == '0':
        a, b = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        if 1500 < a < 1800: print('    %8.1f .. %8.1f  %s' % (a, b, r['Kernel_Name'][:100]))
for sid in ('2', '3'):
    print('  stream %s in the last backward (first 30 dispatches after 1500 us):' % sid)
    n = 0
    for r in rows[i0:]:
        if r.get('Stream_Id') == sid:
            a, b = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
            if a > 1500 and n < 30:
                n += 1
                print('    %8.1f .. %8.1f  %s' % (a, b, r['Kernel_Name'][:110]))
xb = [((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3) for r in rows[i0:] if r.get('Stream_Id') == '3']
print('  chain B (stream 3): %d dispatches, first start %.1f, last end %.1f' % (len(xb), xb[0][0], xb[-1][1]))
# which HW queue / stream each chain's marker kernel ran on
import collections
for sub in ('top_bwd1_kernel', 'gru_bwd1_kernel', 'EpiAtomicAddF32', 'EpiGruZR<unsigned short> >', 'wgrad_kernel', 'maxout_bwd_kernel', 'put_saliency', 'shallow_conv1'):
    c = collections.Counter((r['Queue_Id'], r.get('Stream_Id')) for r in rows[-3000:] if sub in r['Kernel_Name'])
    print('  queue/stream of %-28s %s' % (sub, dict(c)))
# what runs in the tail after the last gru_bwd2
tl = last_end('gru_bwd2_kernel')
for r in rows[i0:]:
    if (int(r['End_Timestamp']) - t0) / 1e3 > tl:
        print('  tail: %-90s %.1f .. %.1f us' % (r['Kernel_Name'][:90], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
P
rm -rf $O/t
