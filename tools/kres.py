"""Registers / scratch / LDS of the kernels of one source file as the compiler reports them (hipcc cross-compiles: no GPU needed).
usage: python tools/kres.py adflow_amd/csrc/kernels_viscous.hip [substring of the mangled name] [-D...]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/kres.o",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for c in rows:
        if pat in c["name"]:
            print(f"{c['name'][:90]:90s} vgpr {c.get('VGPRs', '?'):>4} agpr {c.get('AGPRs', '?'):>3} sgpr {c.get('SGPRs', '?'):>3} "
                  f"scratch {c.get('ScratchSize [bytes/lane]', '?'):>5} occ {c.get('Occupancy [waves/SIMD]', '?')} lds {c.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
