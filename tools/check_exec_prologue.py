#!/usr/bin/env python
"""Static check of the compiler's gfx950 assembly for ONE code-generation defect that was caught in this tree (round 5,
DESIGN section 1 row f1): a lane-wise instruction placed at the top of a reconvergence block BEFORE the `s_or_b64 exec, exec,
s[..]` that re-enables the lanes which skipped the preceding `if`.

    join_block:
        v_accvgpr_write_b32 a76, v51      <- the register allocator's copy of a value that is live in ALL lanes ...
        s_or_b64 exec, exec, s[16:17]     <- ... but the lanes that sat out the `if` are only switched back on HERE

LLVM (roc-7.2.0, clang 22) puts live-range-split copies at "the first non-prologue instruction" of a block; when an unrelated
scalar instruction precedes the exec restore, that point is the top of the block, and the copy runs under the narrowed mask:
the other lanes keep whatever the destination register held.  In train_fast_kernel<2,1,3> this lost the hidden-unit index of
16 lanes, so their weight-gradient stores went to wrong rows -- but only in builds whose register pressure made the allocator
split at that point (six unrelated source lines decided it).  The structured control flow the backend emits has no legitimate
reason to run a vector instruction between a block's start and its exec restore, so any such instruction is reported.

    python tools/check_exec_prologue.py file.s [...]      exit code 1 if anything is found

Lane-independent instructions (v_readlane / v_writelane: SGPR spill traffic, s_*: scalar) are fine there."""
import re
import sys

VEC = re.compile(r"^\s+(v_|ds_|global_|flat_|buffer_|scratch_|image_|tbuffer_)")
LANE_FREE = re.compile(r"^\s+(v_readlane_b32|v_writelane_b32)\b")
EXEC_RESTORE = re.compile(r"^\s+s_or_b64\s+exec,\s*exec,\s*s\[\d+:\d+\]")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
MBB = re.compile(r"^; %bb\.\d+:")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):")
SKIP_BRANCH = re.compile(r"^\s+s_cbranch_execz\s+(\.LBB\d+_\d+)")
# instructions after which a new block is the BODY of an `if` (it runs under the mask just narrowed), not a reconvergence point
NARROWS = re.compile(r"^\s+(s_and_saveexec|s_or_saveexec|s_andn2_saveexec|s_xor_b64\s+(exec|s\[\d+:\d+\],\s*exec)|s_mov_b64\s+exec|"
                     r"s_and_b64\s+exec|s_andn2_b64\s+exec|s_cbranch|s_branch)")
ENDS_SCAN = re.compile(r"^\s+(s_cbranch|s_branch|s_endpgm|s_setpc|s_and_saveexec|s_or_saveexec|s_andn2_saveexec|s_xor_b64\s+exec|"
                       r"s_mov_b64\s+exec|s_and_b64\s+exec|s_andn2_b64\s+exec|s_barrier)")


def check(path):
    """A block is a reconvergence (join) block when its label is the target of an `s_cbranch_execz` (the branch that skips an
    `if` body nobody takes), or when it is entered by falling out of ordinary code (the body of an `if` that has no skip branch).
    In a join block every vector instruction before the first `s_or_b64 exec, exec, s[..]` is reported."""
    lines = open(path, errors="replace").read().split("\n")
    skip_targets = set()
    func = "?"
    for line in lines:
        m = FUNC.match(line)
        if m and not line.startswith(".L"):
            func = m.group(1)
        m = SKIP_BRANCH.match(line.split(";")[0])
        if m:
            skip_targets.add((func, m.group(1)))
    found = []
    func = "?"
    pending = None            # vector instructions since the start of the current JOIN block; None = not in one / scan over
    prev_code = ""
    for no, line in enumerate(lines, 1):
        m = FUNC.match(line)
        if m and not line.startswith(".L"):
            func, pending, prev_code = m.group(1), None, ""
            continue
        lab = LABEL.match(line)
        if lab or MBB.match(line):
            is_skip_target = bool(lab) and (func, lab.group(1)) in skip_targets
            falls_in = bool(prev_code) and not NARROWS.match(prev_code)
            pending = [] if (is_skip_target or (falls_in and not lab)) else None
            continue
        code = line.split(";")[0]
        if not code.strip() or code.lstrip().startswith("."):
            continue
        prev_code = code
        if pending is None:
            continue
        if EXEC_RESTORE.match(code):
            for (n, l) in pending:
                found.append((path, func, n, l.strip(), no, code.strip()))
            pending = None
        elif ENDS_SCAN.match(code):
            pending = None
        elif VEC.match(code) and not LANE_FREE.match(code):
            pending.append((no, code))
    return found


def main(argv):
    bad = []
    for p in argv:
        bad += check(p)
    for (path, func, n, ins, m, res) in bad:
        print("%s:%d: in %s: `%s` runs before the exec restore `%s` (line %d)" % (path, n, func, ins, res, m))
    print("check_exec_prologue: %d file(s), %d finding(s)" % (len(argv), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
