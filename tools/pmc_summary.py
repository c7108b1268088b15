"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per launch, per kernel (k4_* only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            if 'k4_' not in name and not name.startswith('k_'):
                continue
            acc[name.split('(')[0]][row['Counter_Name']].append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
for k in sorted(acc):
    print(f'## {k}\n\n| counter | mean per launch | launches |\n|---|---|---|')
    for c in sorted(acc[k]):
        per = defaultdict(float)
        for did, v in acc[k][c]:
            per[did] += v                       # one row per XCD/instance: sum within a dispatch
        vals = list(per.values())
        print(f'| {c} | {sum(vals) / len(vals):.4g} | {len(vals)} |')
    print()
