"""Prints the per-kernel rows of a rocprofv3 --kernel-trace --stats output directory: name, calls, average ns, share."""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        print("%-90s %6s %14.1f %7s" % (r['Name'][:90], r['Calls'], float(r['AverageNs']), r['Percentage']))
