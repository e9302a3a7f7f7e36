"""Summarise a rocprofv3 --pmc CSV: mean counter value per dispatch for our kernels."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'k_bounce' not in k and 'k_intersect' not in k and 'k_walk' not in k: continue
            k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print(d, k, ' '.join(f'{c}={sum(x)/len(x):.4g}' for c, x in sorted(v.items())))
