"""Aggregate a SVMC_RECORD_TOLERANCES recording (tests/conftest.py) per assert site: the worst fraction of the allowed
tolerance that was used (1 = at the bound) and the worst plain relative deviation.  usage: summarise_tolerances.py <in> <out>"""
import collections
import re
import sys

acc = collections.OrderedDict()
for ln in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+) rtol=(\S+) atol=(\S+) used=(\S+) max_rel=(\S+) n=(\d+)", ln)
    if not m:
        continue
    key = (m.group(1), m.group(2), m.group(3))
    a = acc.setdefault(key, [0.0, 0.0, 0, set()])
    a[0], a[1], a[2] = max(a[0], float(m.group(5))), max(a[1], float(m.group(6))), a[2] + 1
    a[3].add(m.group(4))
with open(sys.argv[2], "w") as fh:
    fh.write("# observed deviations at every np.testing.assert_allclose of `pytest tests/test_gpu_parity.py -m gpu` on an MI355X\n"
             "# used = max |actual-desired| / (atol + rtol |desired|)   (1 = at the tolerance);  max_rel = max |actual-desired| / |desired|\n"
             "# (max_rel is meaningless where desired holds exact zeros: those sites are absolute-tolerance checks)\n")
    for (site, fn, rt), (u, r, c, at) in acc.items():
        ats = sorted(at, key=float)
        atol = ats[0] if len(ats) == 1 else f"{ats[0]}..{ats[-1]}"
        rel = f"{r:9.2e}" if r < 1e100 else "   (zeros)"
        fh.write(f"{site:24s} {fn[:52]:52s} rtol={rt:7s} atol={atol:22s} calls={c:4d} used={u:9.2e} max_rel={rel}\n")
