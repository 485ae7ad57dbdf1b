import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print("%-62s calls %5s total_ms %9.3f avg_us %9.1f pct %6s" % (r['Name'][:62], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
