"""Text summary of rocprofv3 result databases (rocpd sqlite, the default output of ROCm 7):
kernel statistics (the `top_kernels` view = --stats) and, when counters were collected, the mean value of every
counter per dispatch and kernel.   usage: rocpd_summary.py <results.db | dir> [...]"""
import glob, os, re, sqlite3, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)\(", name)
    return m.group(1) if m else name[:70]


for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".db") else sorted(glob.glob(os.path.join(arg, "**", "*.db"), recursive=True))
    for f in files:
        c = sqlite3.connect(f)
        print("==", f)
        rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        if rows:
            print("  %-56s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
            for name, calls, tot, avg, pct in rows:
                print("  %-56s %6d %14.1f %14.1f %7.2f" % (short(name)[:56], calls, tot, avg, pct))
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "group by kernel_name, counter_name").fetchall()
        for name, counter, nd, val in rows:
            print("  %-56s %-20s dispatches=%-4d mean=%.6g" % (short(name)[:56], counter, nd, val))
