#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls,
total/average/min/max duration, registers, LDS, scratch.  Writes markdown."""
import sqlite3
import sys


def main(db, out=None, title=""):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(workgroup_x)||'x'||max(workgroup_y)||'x'||max(workgroup_z) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary {title}", "",
             "| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | wg |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|"]
    for r in rows:
        lines.append(f"| `{r[0][:90]}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
                     f"{100.0 * r[2] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
