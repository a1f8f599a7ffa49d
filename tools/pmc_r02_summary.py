"""Summarises the counter passes of tools/gpu_pmc_r02.sh: per kernel mean counter value per launch."""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]


def short(name):
    m = re.search(r"(dest_kernel|bin_kernel|combine_kernel|quad_backward_shared_kernel|quad_forward_kernel|"
                  r"wgrad_kernel<\d, \d>|expand_kernel<[^>]*>|reduce_partials<[^>]*>|forward_kernel<\w+>|backward_kernel<\w+>)", name)
    if m:
        return m.group(1)
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        mt = re.search(r"MT\d+x\d+x\d+", name)
        return ("Custom_" if name.startswith("Custom") else "") + name[:14] + "_" + (mt.group(0) if mt else "")
    return None


def collect(pattern):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{pattern}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if k:
                res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res


print("== HBM traffic of the MSDA kernels (bytes per launch; FETCH_SIZE doubled: gfx950 counts 128-B requests as 64 B)")
traffic = {}
for grp in ("msda", "fwd"):
    res = collect(f"{grp}_*")
    for k, d in sorted(res.items()):
        f = d.get("FETCH_SIZE", [])
        w = d.get("WRITE_SIZE", [])
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
        fb = 2 * 1024 * sum(f) / max(1, len(f))
        wb = 1024 * sum(w) / max(1, len(w))
        traffic[k] = {"fetch_bytes": fb, "write_bytes": wb, "launches": max(len(f), len(w))}
        print(f"  {k:34s} fetch {fb / 1e6:9.1f} MB  write {wb / 1e6:9.1f} MB  (x{max(len(f), len(w))})")
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)

print("== MFMA utilisation of the dense kernels (one eager train step x3; per launch means)")
res = collect("mfma_g*")
rows = []
for k, d in res.items():
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    gui = d.get("GRBM_GUI_ACTIVE", [])
    if not busy or not gui:
        continue
    n = len(busy)
    mb, g = sum(busy) / n, sum(gui) / len(gui)
    sqb = sum(d.get("SQ_BUSY_CYCLES", [0])) / max(1, len(d.get("SQ_BUSY_CYCLES", [0])))
    mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])
    mo = sum(mops) / max(1, len(mops))
    # MfmaUtil = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs).  GRBM_GUI_ACTIVE comes back summed over the
    # 8 XCDs (expand_kernel: 3.21 M against 215 us of kernel time = 401 k cycles at the ~1.9 GHz a profiled pass runs at),
    # so kernel cycles = GUI_ACTIVE / 8.  Cross-check: expand_kernel issues 93.2 GFLOP / 32768 = 2.84 M MFMAs of 32 cycles
    # = 91.0 M busy cycles, the counter reads 91.2 M.
    util = 100.0 * mb / (g / 8 * 256 * 4) if g else 0.0
    rows.append((mb * n, k, n, mb, g, util, mo, sqb))
for _, k, n, mb, g, util, mo, sqb in sorted(rows, reverse=True)[:14]:
    print(f"  {k:44s} x{n:4d}  MFMA_BUSY {mb:12.0f}  GUI_ACTIVE {g:10.0f}  MfmaUtil {util:5.1f} %  MOPS_BF16 {mo:12.0f}")
