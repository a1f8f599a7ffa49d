"""Instruction mix of a kernel's loops, read from the device assembly: what one iteration of every loop ISSUES, by pipe
(plain VALU, slow VALU = DPP / packed / conversions / dot2, SALU, LDS, vector memory, scratch, matrix) -- the static side
of "this kernel is VALU-issue bound": the per-sample / per-step instruction counts that DESIGN.md quotes, taken from the code
the compiler actually produced instead of from the source.  Uses LLVM's own loop annotations in the assembly (`Loop Header:
Depth=`, `in Loop: Header=`), so a block belongs to its innermost loop; `--hot` prints only loops that contain a matrix
instruction (the compute loops of the MSDA kernels).

Issue-cycle weights per wave-instruction at 4 waves per SIMD, measured on MI355X (profiles/r03_probe_mfma_tr_rates.txt):
plain VALU 3.0, DPP / v_pk / v_cvt_pk / v_dot2 4.5, v_mfma 4x4x4 8.7, 16x16x32 17.7, 32x32x16 ~34 (2 x 16x16x32's work).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s; python tools/isa_loops.py x.s [filter] [--hot]
"""
import collections
import re
import subprocess
import sys

W = {"valu": 3.0, "valu_slow": 4.5, "mfma4": 8.7, "mfma16": 17.7, "mfma32": 34.0, "salu": 1.0, "lds": 1.0, "vmem": 1.0,
     "scratch": 1.0, "smem": 1.0, "wait": 0.0, "barrier": 0.0, "branch": 1.0}
SLOW = re.compile(r"^(v_pk_|v_cvt_pk|v_dot2|v_mov_b32_dpp|v_permlane|v_readlane|v_writelane|v_readfirstlane)|\b(quad_perm|row_shr|row_shl|row_ror|"
                  r"row_bcast|wave_shr|row_mirror|row_half_mirror|dpp8)\b")


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma"):
        if "4x4x4" in op:
            return "mfma4"
        if "32x32" in op:
            return "mfma32"
        return "mfma16"
    if op.startswith("v_"):
        return "valu_slow" if SLOW.search(t) else "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "salu"


def kernels(path):
    out, cur, name = [], None, None
    for l in open(path):
        l = l.rstrip("\n")
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
            else:
                cur.append(l)
    return out


def loops_of(body):
    """{loop header label: {"depth", "parent", "mix": Counter over the blocks whose innermost loop it is}} + the mix outside loops"""
    loops = collections.OrderedDict()
    outside = collections.Counter()
    cur_loop = None                                        # innermost loop header of the current block
    label = None
    pending_label = None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label = m.group(1)[1:]                         # "LBB0_3" -> matches "BB0_3" in the annotations
            pending_label = label
            cur_loop = None
            c = l.split(";", 1)[1] if ";" in l else ""
            l = "\t;" + c
        t = l.strip()
        if t.startswith(";"):
            hm = re.search(r"Loop Header: Depth=(\d+)", t)
            if hm and pending_label:
                key = pending_label[1:]
                loops.setdefault(key, {"depth": int(hm.group(1)), "parent": None, "mix": collections.Counter(), "n": 0})
                loops[key]["depth"] = int(hm.group(1))
                cur_loop = key
            im = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", t)
            if im and pending_label:
                # annotations list the innermost loop first
                if cur_loop is None:
                    cur_loop = im.group(1)
                    loops.setdefault(cur_loop, {"depth": int(im.group(2)), "parent": None, "mix": collections.Counter(), "n": 0})
                elif loops.get(cur_loop, {}).get("parent") is None and im.group(1) != cur_loop:
                    loops[cur_loop]["parent"] = im.group(1)
            continue
        if not t or t.startswith("."):
            continue
        pending_label = None
        t = t.split(";")[0].strip()
        if not t:
            continue
        k = classify(t)
        if cur_loop is None:
            outside[k] += 1
        else:
            loops[cur_loop]["mix"][k] += 1
            loops[cur_loop]["n"] += 1
    return loops, outside


def cycles(mix):
    return sum(W[k] * n for k, n in mix.items())


def fmt(mix):
    order = ["valu", "valu_slow", "salu", "lds", "vmem", "scratch", "smem", "mfma4", "mfma16", "mfma32", "wait", "barrier", "branch"]
    return "  ".join(f"{k}={mix[k]}" for k in order if mix.get(k))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    hot = "--hot" in sys.argv
    path = args[0]
    flt = args[1] if len(args) > 1 else ""
    for name, body in kernels(path):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem:
            continue
        loops, outside = loops_of(body)
        total = collections.Counter(outside)
        for lp in loops.values():
            total.update(lp["mix"])
        print(f"== {dem[:140]}")
        print(f"   whole kernel: {sum(total.values())} instructions   {fmt(total)}")
        print(f"   outside loops: {fmt(outside)}")
        for key, lp in loops.items():
            mf = lp["mix"]["mfma4"] + lp["mix"]["mfma16"] + lp["mix"]["mfma32"]
            if hot and not mf:
                continue
            vis = cycles({k: v for k, v in lp["mix"].items() if k in ("valu", "valu_slow")})
            mis = cycles({k: v for k, v in lp["mix"].items() if k.startswith("mfma")})
            print(f"   loop {key:10s} depth {lp['depth']} parent {lp['parent'] or '-':10s} {lp['n']:5d} instr/iter   {fmt(lp['mix'])}"
                  f"   | VALU issue {vis:.0f} clk, matrix {mis:.0f} clk per iteration and wave")


if __name__ == "__main__":
    main()
