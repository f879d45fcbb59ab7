"""Copy a rocprofv3 *_kernel_stats.csv into profiles/ with kernel names cut to a readable length."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
    r, w = csv.reader(f), csv.writer(g)
    for row in r:
        if row:
            row[0] = row[0][:110]
        w.writerow(row)
