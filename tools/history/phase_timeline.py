"""Analyse the per-workgroup timestamps dumped by IMSEGM_PHASE_PROF=1 IMSEGM_PHASE_DUMP=<file> (profiling aid)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 2, 16)
for v, name in ((1, 'accum sweep (last launch)'), (0, 'final sweep')):
    d = a[:, v, :]
    d = d[d[:, 15] > 0]
    if not len(d):
        continue
    st, en, hw, xcc = d[:, 10], d[:, 11], d[:, 12], d[:, 13] & 0xf
    t0 = st.min()
    st = (st - t0) / 100.0
    en = (en - t0) / 100.0          # microseconds
    print(name, 'workgroups', len(d), 'span %.1f us' % en.max(), 'WG lifetime us p5/p50/p95 %.1f %.1f %.1f'
          % tuple(np.percentile(en - st, [5, 50, 95])))
    cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7) + 256 * xcc
    u, c = np.unique(cu, return_counts=True)
    print('  CUs used', len(u), 'WGs per CU min/mean/max', c.min(), c.mean(), c.max())
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind='stable')]
    conc = np.cumsum(ev[:, 1])
    t = ev[:, 0]
    dt = np.diff(t)
    edges = np.linspace(0, en.max(), 13)
    out = []
    for i in range(12):
        m = (t[:-1] >= edges[i]) & (t[:-1] < edges[i + 1])
        out.append(int((conc[:-1][m] * dt[m]).sum() / max(dt[m].sum(), 1e-9)))
    print('  resident WGs (whole GPU) over 12 time slices:', out, ' (5 per CU = %d)' % (5 * len(u)))
    print('  last start at %.1f us, first end at %.1f us' % (st.max(), en.min()))

# late starters against the rest: where does the lifetime of the last workgroups go?
print()
for v, name in ((1, 'accum sweep'), (0, 'final sweep')):
    d = a[:, v, :]
    d = d[d[:, 15] > 0]
    if not len(d):
        continue
    t0 = d[:, 10].min()
    st = (d[:, 10] - t0) / 100.0
    life = (d[:, 11] - d[:, 10]) / 100.0
    n = d[:, 15].astype(float)
    edges = np.percentile(st, [0, 25, 50, 75, 90, 97, 100])
    print(name, ': start-time buckets (us) -> n, lifetime mean/p95, cycles of wave 0 per phase p0..p7 (per launch), candidates')
    for i in range(len(edges) - 1):
        m = (st >= edges[i]) & (st <= edges[i + 1] if i == len(edges) - 2 else st < edges[i + 1])
        if not m.any():
            continue
        ph = (d[m, :8] / n[m, None]).mean(0)
        print('  %5.1f-%5.1f  n=%5d life %.1f / %.1f   ' % (edges[i], edges[i + 1], m.sum(), life[m].mean(), np.percentile(life[m], 95)),
              ' '.join('%5.0f' % x for x in ph), '  cand %.1f' % (d[m, 9] / n[m]).mean())
    slow = life > np.percentile(life, 95)
    print('  slowest 5%%: start %.1f..%.1f us, phases' % (st[slow].min(), st[slow].max()),
          ' '.join('%5.0f' % x for x in (d[slow, :8] / n[slow, None]).mean(0)))

# would a longest-first dispatch order shorten the launch?  greedy list scheduling of the measured lifetimes on the slots
import heapq
for v, name in ((1, 'accum sweep'),):
    d = a[:, v, :]
    idx = np.nonzero(d[:, 15] > 0)[0]
    d = d[idx]
    if not len(d):
        continue
    life = (d[:, 11] - d[:, 10]) / 100.0
    cand = d[:, 9] / d[:, 15]
    print('lifetime vs candidates of wave 0: corr %.2f; lifetime vs loop cycles corr %.2f' %
          (np.corrcoef(life, cand)[0, 1], np.corrcoef(life, d[:, 2] / d[:, 15])[0, 1]))
    def makespan(order, slots=1280):
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for i in order:
            t = heapq.heappop(h) + life[i]
            end = max(end, t)
            heapq.heappush(h, t)
        return end
    n = len(life)
    print('greedy makespan on 1280 slots: dispatch order %.1f us, longest first %.1f us, by candidates desc %.1f us, ideal %.1f us'
          % (makespan(range(n)), makespan(np.argsort(-life)), makespan(np.argsort(-cand, kind='stable')), life.sum() / 1280))
