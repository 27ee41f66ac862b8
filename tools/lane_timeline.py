#!/usr/bin/env python3
"""Per-stream occupancy of a rocprofv3 --kernel-trace run of bench.py (tools/prof_lanes.sh): for every stream the share of the
traced span in which one of its kernels was running, the mean duration of each kernel, and the gaps between a kernel's end and
the start of the next kernel on the same stream.  usage: lane_timeline.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'orbx' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: (re.search(r'(k_[a-z_]+)', n) or [None, n[:20]])[1]
n = len(rows)
rows = rows[n // 4:]  # steady state
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    by[r['Stream_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
print("span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
for sid, ks in sorted(by.items()):
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    print("stream %s: busy %.1f %% of the span, %d kernels, gaps mean %.1f us max %.1f us" % (sid, 100.0 * busy / (t1 - t0), len(ks),
          sum(gaps) / max(len(gaps), 1) / 1e3, max(gaps) / 1e3 if gaps else 0))
dur = collections.defaultdict(list)
for r in rows:
    dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(dur.items(), key=lambda x: -sum(x[1])):
    v.sort()
    print("  %-22s n %4d mean %8.1f us median %8.1f max %8.1f" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[-1]))
# how many kernels run at once, time-weighted
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
print("kernels running at once (share of time):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# which kernels run together, time-weighted (round 5): a step is short when an issue-bound kernel (k_fast_wave, k_describe_patch) is
# always among them; time in which only latency-bound kernels run is issue capacity left unused
ev2 = []
for r in rows:
    ev2.append((int(r['Start_Timestamp']), 1, short(r['Kernel_Name']))); ev2.append((int(r['End_Timestamp']), 0, short(r['Kernel_Name'])))
ev2.sort(key=lambda x: (x[0], x[1]))
cur2, last2, combo = collections.Counter(), ev2[0][0], collections.Counter()
for t, d, k in ev2:
    key = "+".join(sorted(kk.replace("k_", "") for kk, c in cur2.items() for _ in range(c))) or "(none)"
    combo[key] += t - last2
    last2 = t
    if d:
        cur2[k] += 1
    else:
        cur2[k] -= 1
tot2 = sum(combo.values())
print("kernels running together (share of time, top 16):")
for k, v in combo.most_common(16):
    print("  %-60s %5.1f %%" % (k, 100.0 * v / tot2))
bound = ("fast_wave", "describe_patch")
print("time with no issue-bound kernel running: %.1f %%" % (100.0 * sum(v for k, v in combo.items() if not any(b in k for b in bound)) / tot2))
