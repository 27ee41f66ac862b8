#!/usr/bin/env python3
"""Timeline summary of a `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` run of tools/host_pipeline.py:
busy time of uploads, copies back and kernels, and how much of the uploads' time runs under something else.
usage: host_pipeline_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)[0]
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(kt))]
cps = []
for r in csv.DictReader(open(mt)):
    cps.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?"))))
t_end = max(e for _, e, _ in ker)
t_beg = t_end - (t_end - min(s for s, _, _ in ker)) // 3  # the last third: steady state


def union(iv):
    iv = sorted((max(s, t_beg), min(e, t_end)) for s, e in iv if e > t_beg and s < t_end)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


span = t_end - t_beg
kinds = sorted(set(k for _, _, k in cps))
print("window %.2f ms; kernels busy %.1f %%" % (span / 1e6, 100 * union([(s, e) for s, e, _ in ker]) / span))
for k in kinds:
    iv = [(s, e) for s, e, kk in cps if kk == k]
    n = sum(1 for s, e in iv if e > t_beg and s < t_end)
    print("copies %-24s busy %.1f %% of the window, %d copies, mean %.3f ms" % (k, 100 * union(iv) / span, n, sum(e - s for s, e in iv) / max(len(iv), 1) / 1e6))
big = sorted([(s, e, k) for s, e, k in cps if e - s > 200000 and s > t_beg], key=lambda x: x[0])[:12]
for s, e, k in big:
    print("  %-20s start %.3f ms, %.3f ms" % (k, (s - t_beg) / 1e6, (e - s) / 1e6))
