"""Print the kernel-time summary of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("%-70s %7s %12s %12s %7s" % ("kernel", "calls", "avg us", "total ms", "%"))
for r in list(csv.DictReader(open(f)))[:top]:
    print("%-70s %7s %12.1f %12.2f %7s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
