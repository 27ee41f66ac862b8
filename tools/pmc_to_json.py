#!/usr/bin/env python3
"""rocprofv3 --pmc passes of bench.py (tools/prof_round.sh) -> profiles/rNN_pmc.json: per kernel and per frame, the vector
instruction count, the VALU issue cycles, scalar / LDS instruction counts and the HBM bytes, plus the average launch
duration of the --stats pass.  bench.py reads the newest profiles/r*_pmc.json for its roofline object.

VALU issue cycles = 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2): on gfx950 an instruction that holds its SIMD's issue
port for 4 cycles counts 1 / 0, a 2-cycle one 1 / 0.5 (calibrated with tools/microbench/valu_rate.hip under the same
counters: tools/prof_valu_counters.sh).  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is quoted raw (the guide's x2
correction is calibrated for 16 B/lane streams, these kernels load 4 B/lane)."""
import collections, csv, glob, json, re, sys

def short(n):
    m = re.search(r'(k_[a-z_]+)', n)
    return m.group(1) if m else None

def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + '/*/*_counter_collection.csv') + glob.glob(path + '/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}

def main():
    out_dir, frames, dst, note = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""
    merged = collections.defaultdict(dict)
    for sub in ('sq', 'sq2', 'sq4', 'fetch', 'write'):
        for k, d in counters(out_dir + '/' + sub).items():
            merged[k].update(d)
    dur = {}
    for f in glob.glob(out_dir + '/stats/*/*kernel_stats.csv') + glob.glob(out_dir + '/stats/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            k = short(r['Name'])
            if k:
                dur[k] = float(r['AverageNs']) / 1e3
    per_frame = {}
    for k, d in sorted(merged.items()):
        if 'SQ_INSTS_VALU' not in d:
            continue
        e = {"valu_instr": d['SQ_INSTS_VALU'] / frames,
             "valu_issue_cycles": 4.0 * (d.get('SQ_ACTIVE_INST_VALU', 0) - d.get('SQ_ACTIVE_INST_VALU2', 0)) / frames,
             "salu_instr": d.get('SQ_INSTS_SALU', 0) / frames, "lds_instr": d.get('SQ_INSTS_LDS', 0) / frames,
             "fetch_bytes": d.get('FETCH_SIZE', 0) * 1024.0 / frames, "write_bytes": d.get('WRITE_SIZE', 0) * 1024.0 / frames,
             "waves": d.get('SQ_WAVES', 0) / frames}
        if k in dur:
            e["avg_launch_us_single_stream"] = dur[k]
            e["valu_frac_single_stream"] = e["valu_issue_cycles"] * frames / (dur[k] * 1e-6) / (256 * 4 * 2.4e9)
        per_frame[k] = e
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha16
    from bench import PMC_WORKLOAD
    json.dump({"source": note, "workload": PMC_WORKLOAD, "frames_per_launch": frames, "kernel_sources_sha16": kernel_sources_sha16(),
               "formula": "valu_issue_cycles = 4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2); valu_frac = valu_issue_cycles x frames / "
                          "launch duration / (256 CUs x 4 SIMDs x 2.4 GHz); bytes = FETCH_SIZE / WRITE_SIZE x 1024 (raw)",
               "per_frame": per_frame}, open(dst, 'w'), indent=1)
    for k, e in per_frame.items():
        print("%-22s VALU %8.0f instr %9.0f issue-cycles  SALU %7.0f  LDS %7.0f  fetch %8.0f B write %7.0f B  %7.1f us  valu_frac %.2f"
              % (k, e["valu_instr"], e["valu_issue_cycles"], e["salu_instr"], e["lds_instr"], e["fetch_bytes"], e["write_bytes"],
                 e.get("avg_launch_us_single_stream", 0), e.get("valu_frac_single_stream", 0)))

if __name__ == "__main__":
    main()
