"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass as
MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  Units: the counters are KiB.
gfx950 correction (same guide, section HBM): for wide coalesced streaming reads (16 B per lane) FETCH_SIZE reports
exactly half of the bytes fetched -- the read side is doubled for the streaming GEMV kernels; WRITE_SIZE is
uncalibrated and small here.  Output: JSON {kernel: {launches, fetch_kib_raw, write_kib_raw, hbm_bytes_per_launch}}.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name",
                      (counter,)).fetchall()
    return {re.sub(r"q3a::\(anonymous namespace\)::", "", n): (c, v) for n, c, v in rows}


def main(fetch_db, write_db, out):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for name, (cnt, fk) in f.items():
        short = re.sub(r"^void ", "", name).split("(")[0]
        wk = w.get(name, (0, 0.0))[1]
        streaming = short.startswith("gemv")
        res[short] = {"launches": cnt, "fetch_kib_raw": round(fk, 2), "write_kib_raw": round(wk, 2),
                      "fetch_correction": 2.0 if streaming else 1.0,
                      "hbm_bytes_per_launch": round(((2.0 if streaming else 1.0) * fk + wk) * 1024)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
        print(f"{k[:70]:70s} n={v['launches']:5d} fetch_raw={v['fetch_kib_raw']:12.1f} KiB write_raw={v['write_kib_raw']:10.1f} KiB -> {v['hbm_bytes_per_launch']/1e6:9.3f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:4])
