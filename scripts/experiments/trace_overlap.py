"""kernel_trace.csv of rocprofv3: for the long naive_conv launches, which queue they ran on and what ran concurrently."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
print("columns:", list(rows[0].keys()))
big = [r for r in rows if "naive_conv" in r["Kernel_Name"] and E(r) - S(r) > 1e6]
qs = {}
for r in rows: qs.setdefault((r.get("Queue_Id"), r.get("Stream_Id")), 0); qs[(r.get("Queue_Id"), r.get("Stream_Id"))] += 1
print("kernels per (queue, stream):", qs)
print("long naive_conv launches:", len(big), "on", {(r.get("Queue_Id"), r.get("Stream_Id")) for r in big})
for b in big[:3]:
  over = [r for r in rows if r is not b and S(r) < E(b) and E(r) > S(b)]
  print("naive %.1f ms overlaps %d kernels: %s" % ((E(b) - S(b)) / 1e6, len(over), sorted({r["Kernel_Name"][:40] for r in over})[:6]))
t0, t1 = S(rows[0]), E(rows[-1])
busy = sum(E(r) - S(r) for r in rows)
print("span %.1f ms, summed kernel time %.1f ms" % ((t1 - t0) / 1e6, busy / 1e6))
