#!/usr/bin/env python
"""profiles/r04_config4_pmc.txt from the raw log of `PART=config4 tools/collect_r04.sh` (gpurun_out/r04/config4_pmc.txt):
    python tools/summarise_config4_pmc.py gpurun_out/r04/config4_pmc.txt > profiles/r04_config4_pmc.txt"""
import re
import sys

raw = open(sys.argv[1]).read()
cases, cur = [], None
for ln in raw.split("\n"):
    m = re.match(r"C4 RoughWell \((.*?)\)\s+chains\s+(\d+) d\s+(\d+) Lf\s+(\d+) M\s+(\d+) eps ([\d.]+):\s+([\d.]+) us / proposal\s+"
                 r"([\d.e+]+) steps/s\s+mfma-frac ([\d.]+)\s+accept ([\d.]+)\s+(\S.*)", ln)
    if m:
        cur = {"name": m.group(1), "N": int(m.group(2)), "d": int(m.group(3)), "M": int(m.group(5)), "eps": float(m.group(6)),
               "us": float(m.group(7)), "steps": float(m.group(8)), "frac": float(m.group(9)), "acc": float(m.group(10))}
        cases.append(cur)
        continue
    m = re.match(r'"void l2hmc::(traj_[^"]+)\(l2hmc::KArgs\)",(\d+),(\d+),([\d.]+)', ln)
    if m and cur is not None:
        cur["avg_ns"], cur["kname"] = float(m.group(4)), m.group(1)
        continue
    m = re.search(r"per-dispatch means over (\d+) dispatches: (\{.*\})", ln)
    if m and cur is not None:
        cur.setdefault("ctr", {}).update(eval(m.group(2)))
out = """# Row J (BASELINE.json configs[3]): Rough Well, 16 384 chains, Lf = 10, H = 10 -- rocprofv3 counter passes of the config-4 kernels
# (tools/collect_r04.sh PART=config4 on 1 x MI355X: per case one --kernel-trace --stats run and three --pmc runs -- FETCH_SIZE; WRITE_SIZE;
#  SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- never combined with a trace domain;
#  this table: tools/summarise_config4_pmc.py).
# One kernel family per width: d = 2 traj_small (one dimension per lane), d = 50 traj_tile (one wave per tile, bf16x3 heads), d = 512 traj_wide
# (LDS-resident state).  Both series: `easy` (eta = 0.1, cos(x / eta)) and the reference's own form (eta = 1e-2, cos(x / eta^2): arguments of 1e4 x).
# Step size tuned per case so that the chains move (mean accept in [0.2, 0.9]); 10 proposals per launch.
#
# derived per launch:  HBM bytes = FETCH_SIZE + WRITE_SIZE (KiB, as rocprofv3 reports them; the gfx950 x2 FETCH correction applies to 16-byte
#   streaming reads and is NOT applied here -- these kernels read 4/8-byte rows; uncalibrated either way, see traffic.json's note);
#   algorithmic bytes = 4 N (2 d + M): x in, x_next out, p per proposal (SURVEY 8(d), T-fused);  GB/s = HBM bytes / average kernel duration;
#   matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)  (GUI_ACTIVE sums the 8 XCDs: 128 x it = the 1024 SIMDs' cycles;
#   the bench kernel gives 0.39 by this formula, 0.40 by round 3's 4 x SQ_WAVE_CYCLES, which only holds at one wave per SIMD).
#
# series                 d    kernel                               us/proposal  steps/s    fp32-roof  accept  eps      HBM MB/launch  alg MB  ratio  GB/s   of 8 TB/s  MFMA busy  VALU instr/launch""".split("\n")
for c in cases:
    ct = c["ctr"]
    hb = (ct["FETCH_SIZE"] + ct["WRITE_SIZE"]) * 1024.0
    alg = 4.0 * c["N"] * (2 * c["d"] + c["M"])
    t = c["avg_ns"] * 1e-9
    out.append("  %-22s %3d  %-36s %9.2f  %.3e   %.3f     %.2f   %.5f  %9.2f   %7.2f  %.2f  %6.1f  %.5f    %.3f      %.3e" % (
        c["name"], c["d"], c["kname"], c["us"], c["steps"], c["frac"], c["acc"], c["eps"], hb / 1e6, alg / 1e6, hb / alg,
        hb / t / 1e9, hb / t / 8e12, ct["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * ct["GRBM_GUI_ACTIVE"]), ct["SQ_INSTS_VALU"]))
out += """#
# Reading.  Config 4 is nowhere near the HBM roof at any width: the fused trajectory moves the state once per LAUNCH (x in, x_next out) and
# 4 bytes of p per chain and proposal -- under 10 GB/s, 0.1 % of 8 TB/s, counter bytes within 0.9-1.3 of the algorithmic count (d = 512 writes the
# current state once per proposal: the rejected-chain restore copy of traj_wide; d = 50 since round 4 as well, traj_tile parks its restore copy
# in x_next).  The "rocprof HBM-BW roofline" BASELINE words for this config is therefore the wrong roof for a T-fused kernel: the binding one is
# instruction issue (matrix pipe busy 0.2-0.3, the rest VALU / transcendental issue), which is what `frac` of the fp32-MFMA roof in the table
# states.  The non-easy series costs within a few per cent of the easy one although it executes ~10 % more VALU instructions (the lanes with
# |x| > 1.29 leave the 3-term Cody-Waite range and send their wave through ocml's full range reduction).
# (fp32 vs bf16 state: measured and declined in round 2, profiles/r02_bf16_state_study.txt -- at 0.1 % HBM utilisation there is nothing for a
# narrower state to win.)
#
# ---- raw log ----""".split("\n")
out += ["# " + l for l in raw.split("\n") if l.strip()]
print("\n".join(out))
