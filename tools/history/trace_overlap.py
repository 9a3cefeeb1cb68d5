#!/usr/bin/env python
"""How busy is the device while several images are in flight?  Reads a rocprofv3 `--kernel-trace --output-format csv` file and
reports, over the steady part of the run: the wall span, the time at least one kernel was running, the mean number of kernels
running at once, the gaps between consecutive kernels of one host thread (= one image stream) and the dispatch rate.

    python tools/trace_overlap.py <p_kernel_trace.csv> [skip_fraction]
"""
import collections
import csv
import sys


def main(path, skip=0.4):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48], r.get('Thread_Id', '0'),
                     r.get('Queue_Id', '0'), int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1),
                     int(r.get('Workgroup_Size_X', 1) or 1) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)))
    rows.sort()
    # the steady part: kernels of the worker threads (every thread but the one with the most dispatches, which sets up and checks), without the first `skip`
    # and the last tenth of their span (fill and drain)
    main_thread = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
    print('dispatches per host thread (whole run): %s' % dict(collections.Counter(r[3] for r in rows)))
    workers = [r for r in rows if r[3] != main_thread]
    if len(workers) > 100:
        t0, t1 = workers[0][0], max(r[1] for r in workers)
        lo, hi = t0 + skip * (t1 - t0), t1 - 0.1 * (t1 - t0)
        rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    else:
        t0, t1 = rows[0][0], max(r[1] for r in rows)
        lo = t0 + skip * (t1 - t0)
        rows = [r for r in rows if r[0] >= lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e3
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy, depth, last = 0, 0, t0
    hist = collections.Counter()
    for t, d in ev:
        if depth > 0:
            busy += t - last
        hist[min(depth, 12)] += t - last
        last, depth = t, depth + d
    total = sum(e - s for s, e, *_ in rows) / 1e3
    print('steady window %.1f us, %d dispatches (%.2f us per dispatch, %.0f k dispatches/s)' % (span, len(rows), span / len(rows), len(rows) / span * 1e3))
    print('device busy (>= 1 kernel) %.1f %%, kernels running at once: mean %.2f' % (100 * busy / 1e3 / span, total / span))
    print('time share by number of kernels running: ' + ' '.join('%d:%.0f%%' % (k, 100 * v / 1e3 / span) for k, v in sorted(hist.items())))
    # workgroup-slot occupancy: workgroups x duration against 256 CUs x 8 (a loose bound: 2048 workgroup slots of 256 lanes)
    slots = sum((e - s) / 1e3 * min(g // max(w, 1), 2048) for s, e, n, th, q, g, w in rows)
    print('workgroup x time: %.0f slot-us = %.1f %% of 2048 slots over the window' % (slots, 100 * slots / (2048 * span)))
    per_thread = collections.defaultdict(list)
    for r in rows:
        per_thread[r[3]].append(r)
    gaps, act = [], []
    for th, rs in per_thread.items():
        rs.sort()
        for a, b in zip(rs, rs[1:]):
            gaps.append((b[0] - a[1]) / 1e3)
        act.append(sum(e - s for s, e, *_ in rs) / 1e3 / span)
    gaps.sort()
    if gaps:
        q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))]
        print('%d host threads; gap between consecutive kernels of a thread: p10 %.1f p50 %.1f p90 %.1f p99 %.1f us, mean %.1f' % (
            len(per_thread), q(.1), q(.5), q(.9), q(.99), sum(gaps) / len(gaps)))
        print('share of the window a thread has a kernel running: mean %.2f' % (sum(act) / len(act)))
    queues = collections.Counter(r[4] for r in rows)
    print('hardware queues used: %s' % dict(queues))
    names = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, *_ in rows:
        names[n][0] += 1
        names[n][1] += (e - s) / 1e3
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]:
        print('  %-48s n=%5d avg %7.2f us  %5.1f %% of kernel time' % (n, c, t / c, 100 * t / total))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
