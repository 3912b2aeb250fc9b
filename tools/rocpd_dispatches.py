#!/usr/bin/env python
"""Single dispatches of the boundary-condition / halo / closure kernels of a rocprofv3 kernel trace (rocpd sqlite), in launch
order with their grids: python tools/rocpd_dispatches.py <t_results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
gy = gx.replace('x', 'y') if gx else None
q = f"select name, {gx}, {gy}, duration, start from kernels where name like '%k_bc_%' or name like '%turb_bc%' or name like '%halo_copy%' or name like '%closures%' order by start limit 400"
rows = c.execute(q).fetchall()
# print one evaluation's worth from the middle
mid = len(rows) // 2
t0 = None
for r in rows[mid:mid + 40]:
    if t0 is None: t0 = r[4]
    print(f"{(r[4]-t0)/1e3:9.1f} us  {r[0][:40]:40s} grid {r[1]}x{r[2]}  {r[3]/1e3:7.2f} us")
