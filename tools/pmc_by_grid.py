"""rocprofv3 --pmc counter_collection.csv -> mean counter value per launch, per (kernel, grid size): layers of different size share a kernel."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            if 'k4_' not in name:
                continue
            key = (name.split('(')[0], row.get('Grid_Size', row.get('Grid_Size_X', '')))
            acc[key][row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
for key in sorted(acc, key=lambda k: (k[0], int(k[1] or 0))):
    print(f'## {key[0]}  grid {key[1]}')
    cs = acc[key]
    print(' | '.join(f'{c} {sum(v.values()) / len(v):.4g} (x{len(v)})' for c, v in sorted(cs.items())))
    print()
