"""Turn the rocprofv3 passes of scripts/pmc_bench.sh into a per-launch counter summary of ONE kernel instantiation.

    python scripts/pmc_summarize.py gpurun_out/pmc_bench "<kernel-name substring>" [grid workgroups] > profiles/rNN_pmc_....json

Per launch (mean over the launches of that kernel in the trace; warm-up launches included -- the counters do not depend
on timing): FETCH_SIZE / WRITE_SIZE (KB, raw), HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction of
MI355X_MICROARCH.md section HBM: FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads), L2 hit rate, the SQ
counters and the derived clock / matrix-pipe occupancy."""
import csv, glob, json, os, sys

def counters(d, substr, grid):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        return {}, 0
    acc, n = {}, {}
    for r in csv.DictReader(open(files[0])):
        if substr in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
    return {k: v / n[k] for k, v in acc.items()}, max(n.values()) if n else 0

def duration_us(d, substr, grid):
    files = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    ts = []
    for r in csv.DictReader(open(files[0])):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
        if substr in r["Kernel_Name"] and (grid is None or g == grid):
            ts.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return sum(ts) / len(ts), len(ts), r

root, substr = sys.argv[1], sys.argv[2]
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else None
grid = wgs * 512 if wgs else None          # Grid_Size in the counter CSV is in work-items (512-thread workgroups)
out = {"kernel_substring": substr, "command": "scripts/pmc_bench.sh: rocprofv3 passes over `python3 bench.py --steps 5 --warmup 2 "
       "--no-cpu-baseline --no-opt-in` (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_* GRBM_GUI_ACTIVE, one pass each, --kernel-trace only)"}
f, nf = counters(os.path.join(root, "fetch"), substr, grid)
w, _ = counters(os.path.join(root, "write"), substr, grid)
t, _ = counters(os.path.join(root, "tcc"), substr, grid)
s, _ = counters(os.path.join(root, "sq"), substr, grid)
out["launches_averaged"] = nf
if f and w:
    out["FETCH_SIZE_KB_raw"], out["WRITE_SIZE_KB_raw"] = f["FETCH_SIZE"], w["WRITE_SIZE"]
    out["fetch_correction"] = "x2 (gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while wide coalesced reads are 128-B requests; MI355X_MICROARCH.md, HBM)"
    out["hbm_read_bytes"] = 2 * f["FETCH_SIZE"] * 1024
    out["hbm_write_bytes"] = w["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes"] + out["hbm_write_bytes"]
if t:
    out["l2"] = {"TCC_HIT_sum": t["TCC_HIT_sum"], "TCC_MISS_sum": t["TCC_MISS_sum"],
                 "hit_rate": t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])}
if s:
    dur, nd, _ = duration_us(os.path.join(root, "sq"), substr, grid // 512 * 512 if grid else None)
    s["duration_us_under_profiler"] = dur
    clk = s["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3
    busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
    s["derived"] = {"clock_GHz": round(clk, 3), "mfma_busy_cycles_per_simd": busy,
                    "mfma_busy_fraction_at_that_clock": round(busy / (clk * 1e3 * dur), 4),
                    "wave_cycles_waiting_fraction": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4)}
    out["sq_counters"] = s
print(json.dumps(out, indent=1))
