"""Instruction mix per BASIC BLOCK of a kernel (tools/isa_loops.py's classes): the blocks that contain matrix instructions are the
per-group bodies of the MSDA level loops, so this is where "VALU instructions per sample" of a kernel is read off the device
assembly.  usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics --cuda-device-only -S x.hip -o x.s
                  python tools/isa_blocks.py x.s "<substring of the demangled kernel name>" [--all]"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_loops as I  # noqa: E402


def blocks_of(path, flt):
    """[(demangled kernel name, [(label, Counter of instruction classes, [instruction text])])] of the kernels whose name holds flt"""
    out = []
    for name, body in I.kernels(path):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem:
            continue
        blocks = []
        cur = ["entry", collections.Counter(), []]
        blocks.append(cur)
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = [m.group(1), collections.Counter(), []]
                blocks.append(cur)
                continue
            t = l.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            t = t.split(";")[0].strip()
            if not t:
                continue
            cur[1][I.classify(t)] += 1
            cur[2].append(t)
        out.append((dem, [tuple(b) for b in blocks]))
    return out


def main():
    path, flt = sys.argv[1], sys.argv[2]
    for dem, blocks in blocks_of(path, flt):
        print("==", dem[:130])
        for n, c, ins in blocks:
            if c.get("mfma4") or c.get("mfma16") or c.get("mfma32") or "--all" in sys.argv:
                buf = sum(1 for t in ins if t.startswith("buffer_load"))
                ds = sum(1 for t in ins if t.startswith("ds_read_b128"))
                print(f"  {n:12s} n={sum(c.values()):4d}  {I.fmt(c)} | ds_read_b128={ds} buffer_load={buf}")


if __name__ == "__main__":
    main()
