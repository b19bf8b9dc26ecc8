#!/usr/bin/env python
"""What a rocprofv3 kernel trace (rocpd sqlite, `--kernel-trace`) says about concurrency and gaps, for the last `--tail-ms` of device activity
(the timed part of a run: warm-up and set-up kernels lie before it).

  python scripts/rocpd_overlap.py <results.db> [--tail-ms 200] [--top 14]

Prints (i) busy time (union of all kernel intervals), idle gaps and the sum of kernel durations in that window -- sum > busy means kernels of
different streams ran at the same time --, and (ii) per kernel (short name): calls, total ms, and the share of its run time during which a kernel
of ANOTHER stream was also running, with the partner that overlapped it most."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("mc::", "")
    return name[:60]


def main():
    path = sys.argv[1]
    tail_ms = float(sys.argv[sys.argv.index("--tail-ms") + 1]) if "--tail-ms" in sys.argv else 200.0
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
    t_end = max(r[2] for r in rows)
    rows = [r for r in rows if r[1] >= t_end - tail_ms * 1e6]
    t0 = min(r[1] for r in rows)
    # union of intervals
    busy = 0
    cur_s, cur_e = rows[0][1], rows[0][2]
    gaps = []
    for _, s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(r[2] - r[1] for r in rows)
    span = t_end - t0
    print("window %.2f ms: %d kernels on %d streams; busy (union) %.3f ms = %.1f %% of the window, idle %.3f ms in %d gaps (median gap %.1f us); sum of kernel "
          "durations %.3f ms = %.2f x busy" % (span / 1e6, len(rows), len({r[3] for r in rows}), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps),
                                               (sorted(gaps)[len(gaps) // 2] / 1e3) if gaps else 0.0, total / 1e6, total / max(1, busy)))
    # pairwise overlap with kernels of other streams (sweep)
    ev = sorted(rows, key=lambda r: r[1])
    dur = collections.Counter()
    calls = collections.Counter()
    ovl = collections.Counter()
    partner = collections.defaultdict(collections.Counter)
    active = []   # (end, name, stream, start)
    for name, s, e, st, _ in ev:
        n = short(name)
        dur[n] += e - s
        calls[n] += 1
        active = [a for a in active if a[0] > s]
        for (ae, an, ast, as_) in active:
            if ast == st:
                continue
            o = min(e, ae) - s
            if o > 0:
                partner[n][an] += o
                partner[an][n] += o
        active.append((e, n, st, s))
    # share of a kernel's time covered by ANY other-stream kernel: sweep per kernel instance against the union of the other streams' intervals
    by_stream = collections.defaultdict(list)
    for name, s, e, st, _ in ev:
        by_stream[st].append((s, e))
    for name, s, e, st, _ in ev:
        cov = 0
        segs = sorted((max(s, a), min(e, b)) for k, iv in by_stream.items() if k != st for (a, b) in iv if b > s and a < e)
        ce = s
        for a, b in segs:
            a = max(a, ce)
            if b > a:
                cov += b - a
                ce = b
        ovl[short(name)] += cov
    print("%-60s %6s %9s %9s  %s" % ("kernel", "calls", "total ms", "co-run %", "overlapped most by"))
    for n, d in dur.most_common(top):
        p = partner[n].most_common(1)
        print("%-60s %6d %9.3f %8.1f%%  %s" % (n, calls[n], d / 1e6, 100.0 * ovl[n] / d, ("%s (%.3f ms)" % (p[0][0], p[0][1] / 1e6)) if p else "-"))


if __name__ == "__main__":
    main()
