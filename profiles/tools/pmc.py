import csv, sys, glob, collections
# usage: pmc.py dir [kernel-substring]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_frontend"
for d in sys.argv[1].split(','):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print("%-28s n=%3d  mean=%.6g  last=%.6g" % (k, len(v), sum(v)/len(v), v[-1]))
