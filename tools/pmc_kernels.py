"""Per-kernel averages of every counter in one or more rocprofv3 `--pmc` databases (rocpd SQLite, *_results.db):
    python tools/pmc_kernels.py OUT1/x_results.db [OUT2/y_results.db ...] [--match substring] [--by-grid]
--by-grid keeps dispatches of one kernel with different grid sizes apart (a GEMM kernel serves several shapes: the grid
size names the shape, e.g. 539 x 512 threads = the encoder qkv projection at 32 clips).
Counters are summed over the dimensions rocprofv3 reports (XCD / SE / ...), averaged over the dispatches of a kernel.
Derived lines: SQ shares of SQ_WAVE_CYCLES, LDS bank-conflict rate, MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) when both are present (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"q3a::\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0]


def main(argv):
    match = None
    by_grid = "--by-grid" in argv
    argv = [a for a in argv if a != "--by-grid"]
    if "--match" in argv:
        i = argv.index("--match")
        match = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    data = defaultdict(dict)
    for path in argv:
        db = sqlite3.connect(path)
        # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch, then average over dispatches
        rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(grid_size), max(workgroup_size) "
                          "from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
        acc = defaultdict(list)
        for k, c, _, v, gs, ws in rows:
            name = short(k) + (f"  [grid {int(gs) // max(int(ws), 1)} workgroups]" if by_grid else "")
            acc[(name, c)].append(v)
        for (k, c), vs in acc.items():
            data[k][c] = (sum(vs) / len(vs), len(vs))
    for k in sorted(data, key=lambda k: -data[k].get("SQ_WAVE_CYCLES", data[k].get("GRBM_GUI_ACTIVE", (0, 0)))[0]):
        if match and match not in k:
            continue
        d = data[k]
        n = max(v[1] for v in d.values())
        print(f"{k}   (dispatches: {n})")
        wc = d.get("SQ_WAVE_CYCLES", (0, 0))[0]
        for c in sorted(d):
            v = d[c][0]
            extra = f"  ({v / wc:6.3f} of wave cycles)" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and ("WAIT" in c or "ACTIVE" in c) else ""
            print(f"  {c:32s} {v:16.0f}{extra}")
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE", (0, 0))[0]:
            print(f"  LDS bank-conflict rate           {d['SQ_LDS_BANK_CONFLICT'][0] / d['SQ_LDS_IDX_ACTIVE'][0]:16.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", (0, 0))[0]:
            print(f"  MfmaUtil                         {d['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (d['GRBM_GUI_ACTIVE'][0] / 8 * 1024):16.3f}")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
