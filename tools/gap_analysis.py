"""Post-process a rocprofv3 kernel trace (csv) of bench.py: GPU busy / idle inside the steady-state window,
gap histogram, and the kernels that precede the longest gaps.  usage: gap_analysis.py <kernel_trace.csv> [frac]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# step boundaries: the fused AdamW kernel (csrc/fused_adamw.hip) ends every step
bounds = [r[1] for r in rows if "step_kernel" in r[2]]
ends = sorted(set(bounds))
print("step durations (ms):", " ".join(f"{(b2 - b1) / 1e6:.1f}" for b1, b2 in zip(ends, ends[1:])))
pick = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # analyse the step ending at ends[-pick]
t0, t1 = ends[-pick - 1], ends[-pick]
win = [r for r in rows if t0 <= r[0] < t1]
print(f"analysed step: {(t1 - t0) / 1e6:.2f} ms")
span = win[-1][1] - win[0][0]
busy = 0; cur_s, cur_e = win[0][0], win[0][1]
gaps = []
last_name = win[0][2]
for s, e, n in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
print(f"window {span / 1e6:.2f} ms, {len(win)} kernels; busy (union) {busy / 1e6:.2f} ms = {100 * busy / span:.1f} %; "
      f"idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps; sum of kernel durations {sum(e - s for s, e, _ in win) / 1e6:.2f} ms")
hist = collections.Counter()
tot = collections.Counter()
for g, _, _ in gaps:
    b = 1 if g < 1000 else 2 if g < 2000 else 4 if g < 4000 else 8 if g < 8000 else 16 if g < 16000 else 64 if g < 64000 else 1000
    hist[b] += 1; tot[b] += g
for b in sorted(hist):
    print(f"  gaps < {b:4d} us: {hist[b]:6d}  total {tot[b] / 1e6:7.2f} ms")
by = collections.Counter(); cnt = collections.Counter()
for g, a, b in gaps:
    by[a[:70]] += g; cnt[a[:70]] += 1
print("idle time by preceding kernel:")
for n, t in by.most_common(18):
    print(f"  {t / 1e6:7.2f} ms {cnt[n]:6d}x  avg {t / cnt[n] / 1e3:6.1f} us  {n}")
pairs = collections.Counter(); pcnt = collections.Counter()
for g, a, b in gaps:
    if g >= 12000:
        k = (a[:58], b[:58]); pairs[k] += g; pcnt[k] += 1
print("gaps >= 12 us by (previous kernel -> next kernel):")
for (a, b), t in pairs.most_common(25):
    print(f"  {t / 1e6:6.2f} ms {pcnt[(a, b)]:4d}x  {a}  ->  {b}")
# position of the idle time inside the step (tenths of the step)
pos = [0.0] * 10
for (s0, e0, _), (s1, _, _) in zip(win, win[1:]):
    pass
cur = win[0][1]
for s_, e_, _ in win[1:]:
    if s_ > cur:
        pos[min(9, int(10 * (cur - t0) / (t1 - t0)))] += (s_ - cur) / 1e6
    cur = max(cur, e_)
print("idle ms per tenth of the step:", " ".join(f"{x:.1f}" for x in pos))
print("longest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us  after {a[:60]}  before {b[:60]}")
