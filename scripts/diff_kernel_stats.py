"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (same command, one switch flipped): calls, average and total
per kernel, sorted by the change in total time.  Usage: python scripts/diff_kernel_stats.py A.csv B.csv [iterations]"""
import csv, re, sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"^void ", "", n)[:110]
        out[n] = (int(r["Calls"]), int(r["TotalDurationNs"]) / 1e3)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
it = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort(key=lambda r: -abs(r[0]))
print(f"total A {sum(v[1] for v in a.values()) / it / 1e3:.3f} ms/it, B {sum(v[1] for v in b.values()) / it / 1e3:.3f} ms/it")
for d, k, ca, ta, cb, tb in rows[:40]:
    print(f"{d / it:+9.1f} us/it  A {ca / it:6.1f} x {ta / max(ca, 1):8.1f} us   B {cb / it:6.1f} x {tb / max(cb, 1):8.1f} us   {k}")
