"""Where does the GPU wait for the host?  From a rocprofv3 kernel trace (scripts/prof_bench.sh): the idle time between
consecutive kernels of the last iterations, summed by the kernel that FOLLOWS the gap.  Usage: gap_report.py <kernel_trace.csv> [n_last]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = rows[-n_last:]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
gaps = collections.Counter()
cnt = collections.Counter()
big = []
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 0:
        n = re.sub(r"\(anonymous namespace\)::", "", b["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:60]
        gaps[n] += g
        cnt[n] += 1
        if g > 50:
            big.append((g, re.sub(r"\(anonymous namespace\)::", "", a["Kernel_Name"])[:50], n))
print(f"{len(rows)} kernels: span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms ({(span - busy) / span * 100:.1f} %)")
for n, g in gaps.most_common(15):
    print(f"  {g / 1e3:7.3f} ms idle before {cnt[n]:5d} x {n}   ({g / cnt[n]:.1f} us each)")
big.sort(reverse=True)
print("largest single gaps (us):")
for g, a, b in big[:12]:
    print(f"  {g:8.1f}  after {a}  before {b}")
