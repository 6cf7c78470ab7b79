"""Idle time between kernels in a rocprofv3 --kernel-trace csv."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print("kernels %d span %.3f s busy %.3f s" % (len(rows), span / 1e9, busy / 1e9))
import collections
gap_after = collections.Counter()
cnt = collections.Counter()
dur = collections.Counter()
last_end = rows[0][1]
prev = rows[0][2]
for s, e, n in rows[1:]:
    g = s - last_end
    if g > 0:
        gap_after[(prev, n)] += g
        cnt[(prev, n)] += 1
    dur[n] += e - s
    last_end = max(last_end, e)
    prev = n
for k, v in gap_after.most_common(12):
    print("gap %.3f s over %d transitions (avg %.1f us): %s -> %s" % (v / 1e9, cnt[k], v / 1e3 / cnt[k], k[0], k[1]))
for k, v in dur.most_common(6):
    print("kernel time %.3f s: %s" % (v / 1e9, k))
