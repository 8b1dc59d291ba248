"""Aggregate a rocprofv3 kernel trace over the last `chains` occurrences of a marker kernel:
python scripts/trace_chain.py <kernel_trace.csv> <marker substring> <chains>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
chains = int(sys.argv[3])
sub = rows[idx[-chains]:]
agg = collections.OrderedDict()
for r in sub:
    a = agg.setdefault(r["Kernel_Name"][:56], [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = 0
for k, (c, d) in agg.items():
    print("%-58s calls/chain %.2f  avg %8.1f us  per chain %8.1f us" % (k, c / chains, d / c / 1e3, d / chains / 1e3))
    tot += d
print("kernel time per chain %.1f us; wall span per chain %.1f us" % (
    tot / chains / 1e3, (int(sub[-1]["End_Timestamp"]) - int(sub[0]["Start_Timestamp"])) / chains / 1e3))
