#!/usr/bin/env python3
"""Registers, scratch and the waits of the hottest loop of every kernel of the build, from the compiler's listings
(l2hmc_amd/csrc/build/asm/*.s, written by `make`):

    python tools/kernel_resources.py [--all]          # default: only kernels with scratch or > 256 registers

Round 6's rule: after every change of arithmetic look at `.amdhsa_private_segment_fixed_size` -- the f16x2
commit doubled the fragment registers and the two-tiles-per-wave kernels spilled 220-630 bytes per lane inside their step loops for
half a round.  `resources()` is also what tests/test_abi_cpu.py::test_hot_kernels_do_not_spill reads."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "l2hmc_amd", "csrc", "build", "asm")


def demangle_short(name):
    m = re.match(r"_ZN5l2hmc\d+([A-Za-z_0-9]+?)I(.*)EEv", name)
    if not m:
        m2 = re.match(r"_ZN5l2hmc\d+([A-Za-z_0-9]+?)E", name)
        return m2.group(1) if m2 else name
    args = re.findall(r"L([ib])(\d+)E", m.group(2))
    return "%s<%s>" % (m.group(1), ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args))


def resources(asm_dir=ASM):
    """{file: [(kernel, vgprs incl. AGPRs, scratch bytes, accum_offset)]}"""
    out = {}
    for path in sorted(glob.glob(os.path.join(asm_dir, "*.s"))):
        rows, name, sc, vg = [], None, None, None
        for ln in open(path):
            m = re.match(r"^(_ZN[^:\s]+):", ln)
            if m:
                name = m.group(1)
            elif ".amdhsa_private_segment_fixed_size" in ln:
                sc = int(ln.split()[-1])
            elif ".amdhsa_next_free_vgpr" in ln:
                vg = int(ln.split()[-1])
            elif ".amdhsa_accum_offset" in ln and name is not None:
                rows.append((demangle_short(name), vg, sc, int(ln.split()[-1])))
        out[os.path.basename(path)] = rows
    return out


def main():
    show_all = "--all" in sys.argv
    res = resources()
    if not res:
        sys.exit("no listings under %s: run `make -C l2hmc_amd/csrc` first" % ASM)
    print("%-20s %-52s %6s %8s" % ("file", "kernel", "regs", "scratch"))
    for f, rows in res.items():
        for k, vg, sc, acc in rows:
            if show_all or sc > 0 or vg > 256:
                print("%-20s %-52s %6d %8d%s" % (f, k[:52], vg, sc, "   (AGPRs from %d)" % acc if vg > acc and vg > 256 else ""))


if __name__ == "__main__":
    main()
