#!/usr/bin/env python3
"""Summary of tools/prof_config.sh: per kernel of one BASELINE configuration, the average launch duration (sync call / lanes), VALU
issue cycles per launch (4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2)) and their fraction of the chip's
issue capacity over the launch, and the HBM bytes of the FETCH_SIZE / WRITE_SIZE passes (KiB counters, raw) against 8 TB/s."""
import collections, csv, glob, json, re, sys

out, cfg = sys.argv[1], sys.argv[2]
json_out = sys.argv[3] if len(sys.argv) > 3 else None  # profiles/<tag>_<cfg>_pmc.json: what bench.py's other_configs.<cfg>.roofline reads
PEAK = 256 * 4 * 2.4e9
per_launch = {}


def short(n):
    m = re.search(r'(k_[a-z_]+(?:<[0-9, ]+>)?)', n)
    return m.group(1) if m else None


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def stats(sub):
    d = {}
    for f in glob.glob("%s/%s/*/*kernel_stats.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = short(r['Name'])
            if k:
                d[k] = (float(r['AverageNs']) / 1e3, int(r['Calls']), float(r['Percentage']))
    return d


rate = json.loads(open(out + "/rate.json").read().strip().splitlines()[-1])
print("config %s: %s" % (cfg, json.dumps({k: rate[k] for k in ("frame", "nfeatures", "batch", "window")})))
if "sync" in rate:
    s = rate["sync"]
    print("synchronous call: %.3f ms per batch = %.0f frames/s; algorithmic bytes %.1f MB per frame -> %.0f GB/s = %.3f of 8 TB/s; stages (ms) %s"
          % (s["ms_per_batch"], s["frames_per_s"], s["algorithmic_bytes_per_frame"] / 1e6, s["algorithmic_GBs"], s["algorithmic_frac_of_8TBs"],
             {k: round(v, 3) for k, v in s["stage_ms"].items()}))
if "lanes" in rate:
    l = rate["lanes"]
    print("stream-ordered on %d lanes: %.3f ms per batch = %.0f frames/s; %.0f GB/s algorithmic = %.3f of 8 TB/s"
          % (l["depth"], l["ms_per_batch"], l["frames_per_s"], l["algorithmic_GBs"], l["algorithmic_frac_of_8TBs"]))
if "bf_match" in rate:
    b = rate["bf_match"]
    print("2000 x 2000 descriptor brute-force match: %.4f ms per pair of sets (%d sets per call), %.3g descriptor pairs/s = %.4f of the 4.9 T pairs/s "
          "popcount bound (SURVEY 8(d): 16 lane-ops per pair), nmatches %d" % (b["ms_per_2000x2000"], b["sets_per_call"], b["descriptor_pairs_per_s"],
                                                                            b["frac_of_4.9T_pairs_per_s"], b["nmatches"]))
for title, st, subs in (("synchronous call", "stats", ("sq", "sq4", "fetch", "write")), ("brute-force match", "stats_bf", ("sq_bf", "fetch_bf", "write_bf"))):
    sd = stats(st)
    if not sd:
        continue
    merged = collections.defaultdict(dict)
    for sub in subs:
        for k, d in counters(sub).items():
            merged[k].update(d)
    lanes = stats("stats_lanes") if st == "stats" else {}
    print("\n%s -- per launch: avg us (share of kernel time) | on the lanes avg us | VALU issue cycles, fraction of 256 CUs x 4 SIMDs x 2.4 GHz | "
          "FETCH + WRITE bytes, GB/s, fraction of 8 TB/s" % title)
    for k, (us, calls, pct) in sorted(sd.items(), key=lambda kv: -kv[1][2]):
        d = merged.get(k, {})
        issue = 4.0 * (d.get('SQ_ACTIVE_INST_VALU', 0) - d.get('SQ_ACTIVE_INST_VALU2', 0))
        byts = (d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024.0
        per_launch.setdefault(title, {})[k] = {"avg_us": us, "calls": calls, "avg_us_on_lanes": lanes[k][0] if k in lanes else None,
                                               "valu_issue_cycles": issue, "valu_instr": d.get('SQ_INSTS_VALU'),
                                               "fetch_bytes": d.get('FETCH_SIZE', 0) * 1024.0, "write_bytes": d.get('WRITE_SIZE', 0) * 1024.0}
        print("  %-28s %9.1f us (%4.1f %%, %4d calls) | %9s | %12.0f cycles, valu_frac %.2f | %7.1f MB, %6.0f GB/s, %.3f"
              % (k, us, pct, calls, ("%.1f" % lanes[k][0]) if k in lanes else "-", issue, issue / (us * 1e-6) / PEAK if us else 0,
                 byts / 1e6, byts / (us * 1e-6) / 1e9 if us else 0, byts / (us * 1e-6) / 8e12 if us else 0))

if json_out:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha16
    launch = rate.get("sync", {}).get("launch", {})
    json.dump({"workload": "bench_config %s: %s" % (cfg, json.dumps({k: rate[k] for k in ("frame", "nfeatures", "batch", "window")})),
               "source": "tools/prof_config.sh: rocprofv3 --kernel-trace --stats (durations) and separate --pmc passes (SQ counters, "
                         "SQ_ACTIVE_INST_VALU / VALU2, FETCH_SIZE, WRITE_SIZE) of `python3 tools/bench_config.py --config %s --mode sync`; "
                         "per LAUNCH (a synchronous call issues every kernel once per half batch)" % cfg,
               "kernel_sources_sha16": kernel_sources_sha16(), "frames_per_launch": launch.get("frames_per_launch"),
               "per_launch": per_launch.get("synchronous call", {}), "per_launch_bf_match": per_launch.get("brute-force match", {})},
              open(json_out, "w"), indent=1)
