"""Instruction counts of the main loop of a gfx950 kernel from the compiler's assembly (hipcc -save-temps).

  python tools/isa_count.py <file.s> <regex of the mangled kernel name> [--all-loops]

The "main loop" is the backward branch whose body holds the most instructions.  VALU instructions issue over 4 cycles per
64-wide wavefront on a CDNA SIMD (16 lanes per cycle; FP64 FMA / MUL / ADD at the same rate as 32-bit ops on gfx950, the
transcendental f64 seeds at a quarter of it), so  valu * 4 + trans64 * 12  cycles per wave and loop trip is the issue time
the roofline object of bench.py prices (`roofline.fp64_issue`).
"""
import collections
import re
import sys


def kernel_lines(txt, pat):
    start = None
    for i, l in enumerate(txt):
        if re.match(r"^_Z[^ ]*:", l) and re.search(pat, l):
            start = i
            break
    if start is None:
        return None, None
    end = len(txt)
    for i in range(start, len(txt)):
        if ".end_amdhsa_kernel" in txt[i] or txt[i].startswith(".Lfunc_end"):
            end = i
            break
    return start, end


def classify(counter):
    c = counter
    f64 = sum(v for k, v in c.items() if k.startswith("v_") and "f64" in k and not k.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_cmp", "v_cvt")))
    tr64 = sum(v for k, v in c.items() if k.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    dpp = sum(v for k, v in c.items() if "dpp" in k)
    gl = sum(v for k, v in c.items() if k.startswith(("global_load", "flat_load", "buffer_load")))
    gs = sum(v for k, v in c.items() if k.startswith(("global_store", "flat_store", "buffer_store")))
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    sc = sum(v for k, v in c.items() if k.startswith("scratch_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_barrier")))
    wait = sum(v for k, v in c.items() if k.startswith("s_waitcnt"))
    bar = c.get("s_barrier", 0)
    return dict(total=sum(c.values()), valu=valu, f64=f64, trans64=tr64, dpp=dpp, gload=gl, gstore=gs, ds=ds, scratch=sc,
                salu=salu, waitcnt=wait, barrier=bar, issue_cycles=4 * (valu - tr64) + 16 * tr64)


def loops(txt, start, end):
    label_at = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB[0-9_]+):", txt[i])
        if m:
            label_at[m.group(1)] = i
    out = []
    for i in range(start, end):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", txt[i]) or re.match(r"^\s+s_branch\s+(\.LBB[0-9_]+)", txt[i])
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            a = label_at[m.group(1)]
            c = collections.Counter()
            for l in txt[a:i + 1]:
                mm = re.match(r"^\s+([a-z_0-9]+)", l)
                if mm and not mm.group(1).startswith("."):
                    c[mm.group(1)] += 1
            out.append((a, i, c))
    return out


def main_loop(path, pat):
    txt = open(path).read().split("\n")
    s, e = kernel_lines(txt, pat)
    if s is None:
        return None
    ls = loops(txt, s, e)
    if not ls:
        return None
    a, b, c = max(ls, key=lambda t: sum(t[2].values()))
    r = classify(c)
    r["kernel"] = txt[s].rstrip(":")
    for l in txt[e:e + 60]:
        for key in ("NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs"):
            m = re.search(r";\s*" + key + r":\s*(\d+)", l)
            if m:
                r[key] = int(m.group(1))
    return r


if __name__ == "__main__":
    path, pat = sys.argv[1], sys.argv[2]
    if "--all-loops" in sys.argv:
        txt = open(path).read().split("\n")
        s, e = kernel_lines(txt, pat)
        for a, b, c in loops(txt, s, e):
            print(a - s, b - s, classify(c))
    else:
        print(main_loop(path, pat))
