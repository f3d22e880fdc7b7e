"""Average rocprofv3 --pmc counters per kernel over a directory of csv outputs.  usage: pmc_summary.py DIR [substr]"""
import collections, csv, glob, os, sys
root = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sub in k:
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[1:] if len(v) > 1 else v       # drop the cold first dispatch
        print("   %-28s n=%-3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
