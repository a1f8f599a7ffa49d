"""Static scan of a kernel's device assembly for memory waits that serialise what the source meant to overlap: prints, per
kernel, the order of vector-memory instructions, barriers and s_waitcnt vmcnt(N) -- a load followed at once by vmcnt(0)
before the next load is a dependent chain (the compiler folding a select of scalars into an indexed vector load did that
to cell_backward_kernel: tools found it, not the profiler).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s; python tools/isa_scan.py x.s [name filter]"""
import re
import subprocess
import sys


def kernels(path):
    lines = open(path).read().split("\n")
    out, cur, name = [], None, None
    for l in lines:
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
            else:
                cur.append(l)
    return out


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(path):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem:
            continue
        ins = [l for l in body if l.startswith("\t") and not l.startswith("\t.")]
        print(f"== {dem[:150]}  ({len(ins)} instructions)")
        prev_load = None
        for i, l in enumerate(body):
            t = l.strip()
            if re.match(r"(global_load|buffer_load|scratch_load|global_store|buffer_store|scratch_store|global_atomic)", t):
                prev_load = i
                print(f"   {i:5d}  {t[:90]}")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                tag = "   <-- waits right after the access above" if prev_load is not None and i - prev_load <= 3 else ""
                print(f"   {i:5d}  {t}{tag}")
            elif t.startswith("s_barrier") or re.match(r"\.LBB\S+:.*Loop Header", t):
                print(f"   {i:5d}  {t[:100]}")


if __name__ == "__main__":
    main()
