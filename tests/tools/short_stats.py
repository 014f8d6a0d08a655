import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    m = re.search(r"(k_[a-z_0-9]+(<\d>)?)", name)
    short = m.group(1) if m else ("rocprim:" + ("sort" if "sort" in name else "scan" if "scan" in name else "other") if "rocprim" in name else name[:40])
    print("%-28s calls %4s total_ms %9.3f avg_us %10.1f min_us %10.1f  %5s%%" % (short, row["Calls"], int(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, int(row["MinNs"]) / 1e3, row["Percentage"]))
