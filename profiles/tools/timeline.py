import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# pick a window in the middle
fe = [r for r in rows if 'k_frontend_wave' in r['Kernel_Name']]
a = int(fe[len(fe)//2]['Start_Timestamp']); b = int(fe[len(fe)//2 + 2]['Start_Timestamp'])
print("window %.1f us" % ((b - a) / 1e3))
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= a and s < b:
        print("%9.1f %9.1f %7.1f q%-3s %s" % ((s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:50]))
