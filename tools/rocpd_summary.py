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
        # the same per kernel over its FULL-SIZE launches only (duration >= half of the kernel's longest dispatch): the average of
        # `top_kernels` mixes in the launches that own no tile (a table / SAD kernel on a ragged assembly exits at once) and the
        # row-block launches of the container path, and is then no figure to hold against bench.py's live kernel time
        try:
            per = {}
            for name, dur in c.execute("select name, duration from kernels"):
                per.setdefault(short(name), []).append(dur / 1000.0)
            big = sorted(((k, [d for d in v if d >= 0.5 * max(v)], len(v)) for k, v in per.items()), key=lambda x: -sum(x[1]))[:12]
            if big:
                print("  full-size launches only (>= half of the kernel's longest):")
                print("  %-56s %6s %14s %14s %14s %9s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "of calls"))
                for k, v, n_all in big:
                    print("  %-56s %6d %14.1f %14.1f %14.1f %9d" % (k[:56], len(v), sum(v) / len(v), min(v), max(v), n_all))
        except sqlite3.Error as exc:
            print("  (no per-dispatch view: %s)" % exc)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "group by kernel_name, counter_name").fetchall()
        for name, counter, nd, val in rows:
            print("  %-56s %-20s dispatches=%-4d mean=%.6g" % (short(name)[:56], counter, nd, val))
