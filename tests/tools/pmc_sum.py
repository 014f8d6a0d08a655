import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_0-9]+(<\d>)?)", row["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    calls[k].add(row["Dispatch_Id"])
want = sys.argv[2:] or ["k_assemble_write", "k_assemble_size", "k_slots_light<1>"]
for k in want:
    if k in agg:
        print(k, "dispatches", len(calls[k]), {c: "%.4g" % v for c, v in sorted(agg[k].items())})
