#!/usr/bin/env python
"""HBM-side traffic per launch of one kernel from two rocprofv3 --pmc passes
(FETCH_SIZE in one, WRITE_SIZE in the other: they do not fit one pass, MI355X_MICROARCH.md
"rocprofv3 PMC slots").  Units and corrections as that guide prescribes for gfx950:
both counters are in KiB; FETCH_SIZE reports half of the bytes of a coalesced streaming
read (TCC_EA0_RDREQ x 64 B while the requests are 128 B) and is doubled; WRITE_SIZE was
calibrated here against the known store volume of the kernel (5 doubles per owned cell)
and needs no correction.

usage: pmc_traffic.py <fetch.db> <write.db> <kernel substring> <workload> <out.json> <source note>"""
import json
import os
import sqlite3
import sys


def avg(db, counter, kern):
    c = sqlite3.connect(db)
    r = c.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?",
                  (counter, "%" + kern + "%")).fetchone()
    return r[0], r[1]


fetch_db, write_db, kern, workload, out, src = sys.argv[1:7]
f, nf = avg(fetch_db, "FETCH_SIZE", kern)
w, nw = avg(write_db, "WRITE_SIZE", kern)
ent = {"kernel": kern, "fetch_size_kib_raw": f, "write_size_kib_raw": w, "launches_averaged": [nf, nw],
       "fetch_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
       "traffic_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
       "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported", "source": src}
tab = {}
if os.path.exists(out):
    tab = json.load(open(out))
tab[workload] = ent
json.dump(tab, open(out, "w"), indent=1)
print(json.dumps(ent))
