"""Adds up the summary lines ("fuzz robots: ..." / "fuzz trees: ...") of fuzz campaign logs.
usage: python tools/fuzz_totals.py <log> [<log> ...]"""
import re, sys
keys = [("cases", r"(\d+) cases stepped"), ("samples", r"\((\d+) samples compared"), ("with contacts", r"(\d+) of them with contacts"),
        ("marginally explained outliers", r"(\d+) counted outliers"), ("within the oracle's own spread", r"(\d+) more beyond a tolerance"),
        ("ill-conditioned", r"(\d+) ill-conditioned"), ("link-pose states", r"link poses of (\d+) states")]
tot = dict.fromkeys((k for k, _ in keys), 0)
runs = failed = 0
for f in sys.argv[1:]:
    for line in open(f, errors="replace"):
        if line.startswith(("fuzz robots:", "fuzz trees:")):
            runs += 1
            for k, pat in keys:
                m = re.search(pat, line)
                tot[k] += int(m.group(1)) if m else 0
            failed += 0 if "no failure" in line else 1
        elif re.match(r"seed \d+: exit", line):
            failed += 1
print(f"TOTAL over {runs} runs: " + ", ".join(f"{v} {k}" for k, v in tot.items()) + f"; runs that stopped on a failure: {failed}")
