"""Per-kernel means of rocprofv3 --pmc passes, one CSV row per (kernel, counter).

Usage: python tools/pmc_summary.py OUT.csv DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --pmc <counter> --output-format csv` pass; every
*counter_collection.csv below it is read.  Output columns:
Kernel_Name, Counter_Name, Dispatches, Mean_Counter_Value_KB  (the counter value as rocprofv3
reports it; FETCH_SIZE and WRITE_SIZE are in KB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    a = acc[(row["Kernel_Name"], row["Counter_Name"])]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Counter_Value_KB"])
        for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
            w.writerow([k, c, n, round(s / n, 1)])
    print(f"{out}: {len(acc)} rows")


if __name__ == "__main__":
    main()
