"""Which HIP API calls is the host inside while the GPU sits in its longest idle gaps?  Joins a rocprofv3
kernel trace with the HIP API trace of the same run.  usage: gap_api.py <kernel_trace.csv> <hip_api_trace.csv>"""
import csv, sys, collections
k = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
k.sort()
api = []
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
api.sort()
bounds = [r[1] for r in k if "step_kernel" in r[2]]
t0, t1 = bounds[-5], bounds[-4]
win = [r for r in k if t0 <= r[0] < t1]
gaps = []
cur = win[0][1]; last = win[0][2]
for s, e, n in win[1:]:
    if s > cur:
        gaps.append((s - cur, cur, s, last, n))
    if e >= cur:
        cur, last = e, n
gaps.sort(reverse=True)
print(f"step {(t1 - t0) / 1e6:.2f} ms")
for g, a, b, pn, nn in gaps[:6]:
    print(f"\ngap {g / 1e3:.1f} us at +{(a - t0) / 1e6:.2f} ms: after {pn[:60]} before {nn[:60]}")
    inside = [(s, e, fn, th) for s, e, fn, th in api if e > a - 200000 and s < b + 50000]
    agg = collections.OrderedDict()
    for s, e, fn, th in inside:
        key = (th, fn)
        agg.setdefault(key, [0, 0, s - a])
        agg[key][0] += 1; agg[key][1] += e - s
    for (th, fn), (c, d, first) in list(agg.items())[:40]:
        print(f"   thread {th[-5:]} {fn:38s} x{c:4d} total {d / 1e3:9.1f} us  (first at {first / 1e3:+9.1f} us rel. gap start)")
