"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir-with-runc/*counter_collection.csv> [...]"""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
            k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        print("==", f)
        for k in agg:
            n = max(1, len(disp[k]))
            print("  %-60s dispatches=%d" % (k, n))
            for c, v in sorted(agg[k].items()):
                print("      %-28s %.6g per dispatch" % (c, v / n))
