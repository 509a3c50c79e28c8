"""Static look at the PGS sweep loops of every step-kernel variant in a -save-temps ISA file: per backward branch
that encloses v_med3 instructions, the number of med3 (rows), VALU, SALU, v_readlane, v_writelane (SGPR spills to VGPR
lanes inside the sweep cost a VALU instruction each) and scratch instructions.
usage: python tools/pgs_loop_isa.py <file.s>"""
import re, sys
asm = open(sys.argv[1]).read().splitlines()
starts = [i for i, l in enumerate(asm) if l.startswith("_ZN2hg14hg_step_kernel") and l.split(":")[0].endswith("E")]
for s in starts:
    e = next(i for i in range(s, len(asm)) if ".end_amdhsa_kernel" in asm[i])
    b = asm[s:e]
    name = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", b[0]).groups()
    lab = {m.group(1): i for i, l in enumerate(b) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(b):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            seg = b[lab[m.group(1)]:i + 1]
            nmed = sum("v_med3" in x for x in seg)
            nval = sum(x.startswith("\tv_") for x in seg)
            # the sweep loop proper: many rows, fewest VALU instructions per row (enclosing loops carry other phases)
            if nmed >= 40 and (best is None or nval / nmed < best[1] / best[0]):
                best = (nmed, sum(x.startswith("\tv_") for x in seg), sum(x.startswith("\ts_") for x in seg),
                        sum("v_readlane" in x for x in seg), sum("v_writelane" in x for x in seg), sum("scratch_" in x for x in seg),
                        sum("s_waitcnt" in x or "s_nop" in x for x in seg))
    print("<%s>" % ",".join(name), "sweep loop: med3 %d  VALU %d (%.2f per row)  SALU %d  readlane %d  writelane %d  scratch %d  waitcnt/nop %d"
          % (best[0], best[1], best[1] / best[0], best[2], best[3], best[4], best[5], best[6]) if best else "no loop found")
