#!/usr/bin/env python
"""Does FETCH_SIZE count reads served by the Infinity Cache?  `rocprofv3 --pmc FETCH_SIZE -- tools/pmc_calib.bin bw2` launches the
same read-only and copy kernels on 1 GiB arrays (HBM) and on 32 MiB arrays (resident in the 256 MB Infinity Cache after the first
pass): the counter per byte read of the two working sets, launch by launch.

usage: pmc_mall.py <fetch.db> [out.json]"""
import json
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select kernel_name, grid_size, value from counters_collection where counter_name='FETCH_SIZE' "
                          "order by dispatch_id"))
    out = {}
    rb_big, rb_small = 16.0 * (1 << 26), 16.0 * (1 << 21)          # bytes read per launch: 1 GiB / 32 MiB of double2
    gbig = (1 << 26) // 8                                           # threads of the 1 GiB launches (8 double2 per thread)
    for key in ("read16u", "copy16u"):
        big = [val for name, g, val in rows if key in name and g == gbig]
        small = [val for name, g, val in rows if key in name and g == gbig // 32]
        if len(big) < 3 or len(small) < 3:
            continue
        fb = sum(big[2:]) / len(big[2:]) * 1024.0                   # (the first two launches are the warm-up)
        fs = sum(small[2:]) / len(small[2:]) * 1024.0
        out[key] = {"launches": [len(big), len(small)], "fetch_bytes_raw_per_byte_read_1gib": fb / rb_big,
                    "fetch_bytes_raw_per_byte_read_32mib": fs / rb_small}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
