#!/usr/bin/env python
"""Gaps between consecutive kernels of the one-clip decode loop, from an un-instrumented rocprofv3 kernel trace:

    (cd /tmp && rocprofv3 --kernel-trace -d DIR -o t -- python bench.py --inner --preset 0.6b --batch 1 --seconds 30 --new-tokens 100 --steps 3 --warmup 1)
    python tools/gap_stats.py DIR

gap = start of kernel i + 1 - end of kernel i, grouped by the pair (kernel i -> kernel i + 1).  The pair argmax_finalize -> first
GEMV of the next token is the seam between two hipGraph launches (one graph per decode step); every other pair is an edge inside
a graph.  Printed per pair: count, median, mean, p10 / p90 (us).  The question it answers: does a step boundary cost more than an
in-graph kernel boundary (it would pay to capture several steps per graph)."""
import glob
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"q3a::\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:60]


def main(d):
    db = glob.glob(d + "/**/*_results.db", recursive=True)[0]
    rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
    pairs = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        pairs.setdefault((short(n0), short(n1)), []).append((s1 - e0) / 1e3)
    print(f"# {len(rows)} dispatches; gap = next start - this end (us)")
    print(f"{'this kernel -> next kernel':124s} {'count':>7s} {'median':>8s} {'mean':>8s} {'p10':>8s} {'p90':>8s}")
    for (a, b), v in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
        if len(v) < 20:
            continue
        v.sort()
        q = lambda f: v[min(len(v) - 1, int(f * len(v)))]
        print(f"{(a + ' -> ' + b):124s} {len(v):7d} {q(0.5):8.2f} {sum(v) / len(v):8.2f} {q(0.1):8.2f} {q(0.9):8.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
