"""scan a rocprofv3 csv result directory for API calls longer than 5 ms; print each with the 6 calls before / after it
(name, start offset from the first record in ms, duration in us) and a histogram of API names."""
import csv
import glob
import sys
from collections import Counter

rows = []
for path in glob.glob(sys.argv[1] + "/**/*_api_trace.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Function", r.get("Name", "?")),
                             r.get("Thread_Id", "")))
            except (KeyError, ValueError):
                pass
rows.sort()
if not rows:
    print("no api trace rows under", sys.argv[1], glob.glob(sys.argv[1] + "/**/*", recursive=True)[:20])
    sys.exit(0)
t0 = rows[0][0]
print(f"{len(rows)} api records, span {(rows[-1][1] - t0) / 1e9:.2f} s")
print("calls by name:", Counter(r[2] for r in rows).most_common(25))
slow = [i for i, r in enumerate(rows) if r[1] - r[0] > 5_000_000]
print(f"{len(slow)} calls longer than 5 ms")
for i in slow:
    s, e, name, tid = rows[i]
    if (s - t0) < 2e9 and "Module" in name:
        continue
    print(f"--- {name} tid {tid} at +{(s - t0) / 1e6:.1f} ms took {(e - s) / 1e3:.0f} us")
    for j in range(max(0, i - 6), min(len(rows), i + 7)):
        s2, e2, n2, t2 = rows[j]
        print(f"   {'>>' if j == i else '  '} +{(s2 - t0) / 1e6:10.3f} ms {(e2 - s2) / 1e3:10.1f} us  {n2} [{t2}]")
