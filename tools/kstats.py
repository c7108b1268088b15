#!/usr/bin/env python3
"""Print the marcher kernels' average durations from rocprofv3 kernel_stats CSVs: tools/kstats.py file.csv [...]"""
import csv, sys
for f in sys.argv[1:]:
    rows = [r for r in csv.reader(open(f)) if r and r[0] != 'Name']
    keep = [r for r in rows if any(k in r[0] for k in ('shade', 'feat_kernel', 'geom3', 'order'))]
    tot = sum(float(r[3]) for r in keep) / 1e3
    print(f'{f.split("/")[-1]:34s} ' + '  '.join(f'{r[0].split("(")[0].replace("void ", "")[:28]} {float(r[3]) / 1e3:7.1f}' for r in keep) + f'   sum {tot:7.1f} us')
