"""Per-iteration kernel time by kernel from a rocprofv3 kernel_stats.csv of the bench command:
python scripts/kstats.py <kernel_stats.csv> <iterations in the process> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
its = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot, out = 0.0, []
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"^void ", "", n)
    t = float(r["TotalDurationNs"]) / 1e3 / its
    tot += t
    out.append((t, float(r["Calls"]) / its, float(r["AverageNs"]) / 1e3, n[:125]))
print(f"total {tot / 1e3:.3f} ms per iteration over {len(rows)} kernels")
fam = {}
for t, c, a, n in out:
    key = ("ring conv" if "ring_kernel" in n else "wgrad split" if "wgrad_split8" in n else "other split conv" if "bf16split_kernel" in n
           else "thin (3-channel)" if "thin" in n else "relayout gy" if "relayout" in n else "slab sums" if ("reduce" in n or "slab_sum" in n)
           else "BatchNorm" if ("bn_" in n or "bn1d" in n or "affine_act" in n or "stats_partial" in n) else "Adam" if "adam" in n
           else "absmax" if "absmax" in n else "pack" if "pack" in n else "vendor GEMM" if ("Cijk" in n or "gemm" in n.lower()) else "other")
    fam[key] = fam.get(key, 0.0) + t
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {v / 1e3:7.3f} ms")
for t, c, a, n in out[:top]:
    print(f"{t:8.1f} us/it {c:6.1f} x {a:7.1f}  {n}")
