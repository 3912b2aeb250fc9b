#!/usr/bin/env python
"""SQ counters per launch of the kernels of one residual evaluation (two rocprofv3 --pmc passes, tools/_gpu_job_sq.sh) -> JSON:
the DYNAMIC wave-instruction counts the FP64-issue roofline of bench.py prefers over the static counts of the compiler's assembly
(both branches of a uniform branch are in the assembly, only one is issued).

usage: pmc_sq.py <a.db> <b.db> <workload> <out.json> <git> <source note>"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import PHASE_OF  # noqa: E402


def main():
    a_db, b_db, workload, out, git, src = sys.argv[1:7]
    ent = {"git": git, "source": src, "kernels": {}}
    for db in (a_db, b_db):
        c = sqlite3.connect(db)
        for name, counter, v, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                             "group by kernel_name, counter_name"):
            phase = next((p for key, p in PHASE_OF if key in name), None)
            if phase is None:
                continue
            e = ent["kernels"].setdefault(phase, {"kernel": name[:100], "launches_averaged": n})
            e[counter] = v
    for e in ent["kernels"].values():
        if e.get("SQ_WAVE_CYCLES"):
            e["valu_active_per_wave_cycle"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"]
            e["wait_any_per_wave_cycle"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    tab = json.load(open(out)) if os.path.exists(out) else {}
    tab[workload] = ent
    json.dump(tab, open(out, "w"), indent=1)
    for p, e in ent["kernels"].items():
        print(f"{p:36s} INSTS_VALU {e.get('SQ_INSTS_VALU', 0):.4g}  VALU active / wave cycle {e.get('valu_active_per_wave_cycle', 0):.3f}  "
              f"waiting / wave cycle {e.get('wait_any_per_wave_cycle', 0):.3f}")


if __name__ == "__main__":
    main()
