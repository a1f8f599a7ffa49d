"""(CPU) Per-kernel HBM traffic (bytes per launch) from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_final_r0{2,3,5}.sh.

usage: pmc_final_summary.py <dir with {msda,fwd,b0}_{FETCH_SIZE,WRITE_SIZE}/> [copy-to.json]
Writes <dir>/traffic.json (and the copy, e.g. profiles/r05_final_traffic.json -- the file bench.py's roofline.traffic reads)."""
import collections
import csv
import glob
import json
import re
import shutil
import sys


def short(name):
    """kernel name of a rocprofv3 row -> key of the traffic table (None: not an MSDA kernel of interest).  REFDIM = 0 is the
    B0-signature instantiation, REFDIM = 2 / 4 carries the module's geometry backward as its epilogue ("+geometry")."""
    m = re.search(r"(patch_dest_kernel|patch_dest_multi_kernel|cell_backward_kernel<[^>]*>|cell_records_backward_kernel<[^>]*>|cell_forward_kernel<[^>]*>|bin2_kernel|"
                  r"dest_kernel|bin_kernel|combine_kernel|quad_backward_shared_kernel<[^>]*>|quad_forward_fused_kernel|"
                  r"quad_forward_kernel)", name)
    if not m:
        return None
    k = m.group(1)
    if k.startswith("cell_backward_kernel"):
        refdim = re.match(r"cell_backward_kernel<\s*(\d+)", k)
        k = "cell_backward_kernel" + ("" if refdim and refdim.group(1) == "0" else "+geometry")
    elif k.startswith("cell_records_backward_kernel"):
        refdim = re.match(r"cell_records_backward_kernel<\s*(\d+)", k)
        k = "cell_records_backward_kernel" + ("" if refdim and refdim.group(1) == "0" else "+geometry")
    elif k.startswith("cell_forward_kernel"):
        k = "cell_forward_kernel" + ("+records" if re.search(r", [123]>", k) else "")
    elif k.startswith("quad_backward_shared_kernel"):
        k = "quad_backward_shared_kernel" + ("+geometry" if re.search(r", [24]>", k) else "")
    return k


def summarise(out):
    res = {}
    for grp in ("msda", "fwd", "b0"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{out}/{grp}_*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
            # rocprofv3 reports both in KB; FETCH_SIZE doubled (gfx950 tallies 128-byte requests as 64 B, MI355X_MICROARCH.md)
            res[f"{grp}:{k}"] = {"fetch_bytes": 2 * 1024 * sum(f) / max(1, len(f)), "write_bytes": 1024 * sum(w) / max(1, len(w)),
                                 "launches": max(len(f), len(w))}
    return res


if __name__ == "__main__":
    out = sys.argv[1]
    res = summarise(out)
    for k, v in sorted(res.items()):
        print(f"{k:50s} fetch {v['fetch_bytes'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB  (x{v['launches']})")
    json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
    if len(sys.argv) > 2 and res:
        shutil.copyfile(f"{out}/traffic.json", sys.argv[2])
