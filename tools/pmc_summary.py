#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, sys, collections, glob, re
def short(n):
    m = re.search(r'(k_[a-z_]+)', n)
    return m.group(1) if m else n[:30]
for path in sys.argv[1:]:
    for f in glob.glob(path + '/*/*_counter_collection.csv') + glob.glob(path + '/*_counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'): continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in sorted(acc):
            print(k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c, v in sorted(acc[k].items())), 'n=%d' % len(next(iter(acc[k].values()))))
