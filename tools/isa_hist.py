#!/usr/bin/env python3
"""Instruction histogram of a region of a gfx950 kernel, from the compiler's assembly.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o ek1.s l2hmc_amd/csrc/traj_ek1.hip
  python tools/isa_hist.py ek1.s --kernel 'traj_kernelILi1ELi1ELi4ELi3E' [--from .LBB6_142 --to .LBB6_153]

Without --from/--to the region is the innermost (deepest) loop of the kernel as annotated by
LLVM ("Inner Loop Header: Depth=k" ... up to the last line tagged with that header).  Prints the
count per instruction class and per mnemonic -- the evidence the VALU-issue work in DESIGN.md is
argued from (profiles/r02_traj_isa_hist.txt).
"""
import argparse
import collections
import re
import sys

CLASSES = [
    ("mfma", r"^v_mfma"),
    ("trans (v_exp/v_rcp/v_log/v_sqrt/v_sin/v_cos)", r"^v_(exp|rcp|log|sqrt|rsq|sin|cos)_"),
    ("valu packed f32 (v_pk_*)", r"^v_pk_"),
    ("valu select (v_cndmask)", r"^v_cndmask"),
    ("valu max/min (relu)", r"^v_(max|min)(3)?_"),
    ("valu fma/mul/add/sub f32 (scalar)", r"^v_(fma|fmac|mul|add|sub|subrev|mad)_f32"),
    ("valu mov / accvgpr", r"^v_(mov|accvgpr)"),
    ("valu cross-lane (dpp/permute/readlane)", r"^v_(readlane|readfirstlane|writelane|permlane|swap)|^ds_(bpermute|permute|swizzle)"),
    ("valu compare", r"^v_cmp"),
    ("valu int / other", r"^v_"),
    ("lds read", r"^ds_read"),
    ("lds write", r"^ds_write"),
    ("vmem load", r"^(global|buffer|flat)_load"),
    ("vmem store", r"^(global|buffer|flat)_store"),
    ("s_waitcnt", r"^s_waitcnt"),
    ("s_barrier", r"^s_barrier"),
    ("s_nop", r"^s_nop"),
    ("branch", r"^s_(cbranch|branch)"),
    ("salu / smem other", r"^s_"),
]


def kernel_lines(path, kernel):
    out, inside = [], False
    pat = re.compile(kernel)
    for ln in open(path):
        if not inside:
            if re.match(r"^[_A-Za-z][\w.$]*:", ln) and pat.search(ln):
                inside = True
            continue
        if ln.strip().startswith(".end_amdhsa_kernel") or ln.strip().startswith(".section"):
            break
        out.append(ln.rstrip("\n"))
    if not out:
        sys.exit("kernel not found: " + kernel)
    return out


def innermost_region(lines):
    # LLVM prints the label on the line before the "=> This Inner Loop Header" comment for nested loops
    best = None
    for i, ln in enumerate(lines):
        m = re.search(r"Inner Loop Header: Depth=(\d+)", ln)
        if not m:
            continue
        d = int(m.group(1))
        j = i
        while j >= 0 and not re.match(r"^(\.LBB\d+_\d+):", lines[j]):
            j -= 1
        lab = re.match(r"^\.L(BB\d+_\d+):", lines[j]).group(1)
        last = j
        for k in range(j, len(lines)):
            if ("Header=" + lab + " ") in lines[k] or ("Header=" + lab) == lines[k].strip()[-len("Header=" + lab):]:
                last = k
        # extend to the end of the last tagged block
        k = last + 1
        while k < len(lines) and not re.match(r"^\.LBB|^; %bb", lines[k]):
            k += 1
        size = k - j
        if best is None or d > best[0] or (d == best[0] and size > best[3]):
            best = (d, j, k, size)
    return best[1], best[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--from", dest="lo")
    ap.add_argument("--to", dest="hi")
    ap.add_argument("--per-step", type=float, default=1.0, help="divide counts (e.g. unrolled iterations)")
    a = ap.parse_args()
    L = kernel_lines(a.asm, a.kernel)
    if a.lo:
        i0 = next(i for i, l in enumerate(L) if l.startswith(a.lo + ":"))
        i1 = next(i for i, l in enumerate(L) if l.startswith(a.hi + ":"))
    else:
        i0, i1 = innermost_region(L)
    cls, mn = collections.Counter(), collections.Counter()
    for ln in L[i0:i1]:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        for name, pat in CLASSES:
            if re.match(pat, op):
                cls[name] += 1
                break
        mn[re.sub(r"_e(32|64)$", "", op)] += 1
    tot = sum(cls.values())
    valu = sum(v for k, v in cls.items() if k.startswith("valu") or k.startswith("trans"))
    print("kernel %s  region lines %d..%d  (%d instructions, %d VALU incl. transcendentals)" % (a.kernel, i0, i1, tot, valu))
    for name, _ in CLASSES:
        if cls[name]:
            print("  %-48s %5d" % (name, cls[name]))
    print("  -- by mnemonic")
    for op, n in mn.most_common():
        print("     %-28s %5d" % (op, n))


if __name__ == "__main__":
    main()
