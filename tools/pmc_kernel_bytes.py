#!/usr/bin/env python
"""HBM-side bytes per launch of EVERY kernel of a profiled command, from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the
other; both counters in KiB; gfx950: FETCH_SIZE x 2, see tools/pmc_traffic.py and profiles/r03_mall_fetch.json).  Markdown table,
largest total first; `cells` turns the per-launch bytes into B per cell.

usage: pmc_kernel_bytes.py <fetch.db> <write.db> <cells> <out.md> <title>"""
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    return {n: (v, k) for n, v, k in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
                                               "group by kernel_name", (counter,))}


def main():
    fdb, wdb, cells, out, title = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    rows = []
    for name in set(f) | set(w):
        if name.startswith("__amd_rocclr") or "at::native" in name:
            continue
        fv, fn = f.get(name, (0.0, 0))
        wv, wn = w.get(name, (0.0, 0))
        rb, wb = 2.0 * fv * 1024.0, wv * 1024.0
        rows.append((name, max(fn, wn), rb, wb, (rb + wb) * max(fn, wn)))
    rows.sort(key=lambda r: -r[4])
    lines = [f"# HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x 2 on gfx950) {title}", "",
             "| kernel | launches | read MB | written MB | B / cell / launch | total GB |", "|---|---:|---:|---:|---:|---:|"]
    for name, n, rb, wb, tot in rows:
        lines.append(f"| `{name[:90]}` | {n} | {rb / 1e6:.1f} | {wb / 1e6:.1f} | {(rb + wb) / cells:.1f} | {tot / 1e9:.2f} |")
    txt = "\n".join(lines) + "\n"
    open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
