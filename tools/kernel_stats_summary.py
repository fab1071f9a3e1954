"""Summary of a rocprofv3 --kernel-trace --stats run: python tools/kernel_stats_summary.py <dir>"""
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[0])))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {files[0]}: {len(rows)} kernels, {total / 1e6:.2f} ms in kernels")
for r in rows:
    name = re.sub(r"nolzss::\(anonymous namespace\)::|nolzss::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    print(f"{name[:70]:70s} calls={int(r['Calls']):6d} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):5.2f}")
