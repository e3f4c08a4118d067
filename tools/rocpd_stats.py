"""Summarise a rocprofv3 rocpd sqlite database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into the
per-kernel table rocprofv3's own --stats CSV would hold: calls, total/avg/min/max duration (us), share."""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                           "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"# source: {path}", f"# total kernel time: {tot:.1f} us over {sum(r[1] for r in rows)} dispatches",
             f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for r in rows:
        n = re.sub(r"q3a::\(anonymous namespace\)::", "", r[0])[:100]
        lines.append(f"{n:100s} {r[1]:7d} {r[2]:12.1f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.2f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
