#!/bin/bash
# Developer probe: kernel trace of a short bench run; prints, for the timed loop, the mean duration of the accumulate and update
# kernels and the mean gaps between them (end of one kernel to start of the next on the same stream).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gap_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-ns --no-coarse --steady 0 > /tmp/gap_bench.json 2>/tmp/gap_bench.err
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/gap_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
acc = [i for i, (n, s, e) in enumerate(seq) if 'icp_accumulate_kernel' in n]
acc = acc[20:]   # skip the warm-up launches
d_acc, d_upd, g1, g2, others = [], [], [], [], {}
for a, b in zip(acc[:-1], acc[1:]):
    n, s, e = seq[a]
    d_acc.append(e - s)
    mids = seq[a + 1:b]
    upd = [m for m in mids if 'icp_update_kernel' in m[0] or 'icp_reduce_update' in m[0]]
    if len(mids) == 1 and upd:
        d_upd.append(upd[0][2] - upd[0][1]); g1.append(upd[0][1] - e); g2.append(seq[b][1] - upd[0][2])
    else:
        for m in mids: others[m[0][:50]] = others.get(m[0][:50], 0) + 1
mean = lambda v: sum(v) / max(len(v), 1) / 1e3
print(f"iterations with only [accumulate, update]: {len(g1)} of {len(acc) - 1}")
print(f"accumulate {mean(d_acc):.1f} us | gap {mean(g1):.1f} us | update {mean(d_upd):.1f} us | gap {mean(g2):.1f} us | sum {mean(d_acc) + mean(g1) + mean(d_upd) + mean(g2):.1f} us")
print("other kernels between accumulates:", others)
PY
