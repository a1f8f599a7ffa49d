"""VALU-issue floor of the encoder's MSDA kernels per N = 4 call, computed from the gfx950 assembly (no GPU): the numbers that
profiles/r06_records_route_static.txt quotes, by command instead of by hand.

    python tools/static_floor.py            # compiles csrc/msda_patch.hip (product + ablation build) to assembly, prints the table

Model (what "if nothing ever stalled" means here): a wave-instruction occupies its SIMD's issue port for the cycles measured in round
3 at 4 waves per SIMD (profiles/r03_probe_mfma_tr_rates.txt: plain VALU 3.0, DPP / packed / conversions 4.5, MFMA 4x4x4 8.7,
16x16x32 17.7; scalar instructions 1.0 -- they share the wave's instruction stream); a kernel's floor = sum over its waves of the
issue cycles of what each wave executes / (1 024 SIMDs x 2.4 GHz).  What a wave executes: the blocks outside loops once, the
level loop 4 times (without the blocks of the route the level does not take: the staged route's blocks read LDS, the direct
route's read through the buffer unit), the staging loop ~5 rounds per kernel at the bench shape (2 for a 28 x 28-pixel level-0
window, 1 for each coarser level), the patch pass's step loop once per step of the census (profiles/r03_patch_census.txt).
It is arithmetic on instruction counts -- an ESTIMATE of a lower bound, not a measurement; the product kernels measured in round 3
run at 2.3-2.9 x this figure."""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_blocks as B  # noqa: E402
import isa_loops as I  # noqa: E402

W = dict(I.W, salu=1.0, smem=1.0, branch=1.0, lds=1.0, vmem=1.0, scratch=1.0)
SIMDS, GHZ = 1024, 2.4
WAVES_CELL = 4 * 8 * 77 * 8                    # N x heads x cells (7 x 11 at 800 x 1333) x 8 waves per workgroup
STEPS = {"init": 253424, "model": 297948}      # patch pass, steps of 32 (group, patch) pairs per N = 4 call (the census)
STAGING_ROUNDS = 5


VALU = ("valu", "valu_slow")
ALL = ("valu", "valu_slow", "salu", "smem", "branch", "lds", "vmem", "scratch")


def issue(mix, keys=VALU):
    """issue cycles of an instruction mix: the vector ALU's (default: what DESIGN.md's floors and the 2.3-2.9 x of the measured kernels
    refer to -- scalar, LDS and memory instructions issue next to another wave's VALU instruction), or of every class (ALL)"""
    return sum(W[k] * mix.get(k, 0) for k in keys)


def assembly(defines):
    out = os.path.join(tempfile.mkdtemp(prefix="static_floor_"), "msda_patch.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
                    "-w", *defines, os.path.join(ROOT, "rlipv2_amd", "csrc", "msda_patch.hip"), "-o", out], check=True, timeout=900)
    return out


def kernel(path, flt):
    for name, body in I.kernels(path):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt in dem:
            return I.loops_of(body), B.blocks_of(path, flt)[0][1]
    raise SystemExit(f"no kernel {flt!r} in {path}")


def cell_kernel_floor(path, flt):
    """records backward / cell forward: outside once + 4 x (level loop minus the direct route's blocks) + staging rounds"""
    (loops, outside), blocks = kernel(path, flt)
    level = max((lp for lp in loops.values() if lp["depth"] == 1), key=lambda lp: lp["n"])
    inner = [lp for k, lp in loops.items() if lp["depth"] == 2 and lp["mix"].get("vmem", 0) >= 4 and not any(
        lp["mix"].get(m) for m in ("mfma4", "mfma16"))]
    direct = collections.Counter()
    for _, c, ins in blocks:                      # the buffer-load route's sample blocks: not executed on a staged level
        if c.get("mfma4") and sum(t.startswith("buffer_load") for t in ins) >= 8:
            direct.update(c)
    def total(keys):
        lvl = issue(level["mix"], keys) - issue(direct, keys)
        stg = issue(inner[0]["mix"], keys) if inner else 0.0
        return issue(outside, keys) + 4 * lvl + STAGING_ROUNDS * stg, lvl, stg
    per_wave, lvl, stg = total(VALU)
    mfma = 4 * (level["mix"].get("mfma4", 0) - direct.get("mfma4", 0)) * I.W["mfma4"]
    to_us = WAVES_CELL / SIMDS / GHZ / 1e3
    return {"per_wave_clk": per_wave, "level_clk": lvl, "staging_round_clk": stg, "outside_clk": issue(outside),
            "us": per_wave * to_us, "mfma_us": mfma * to_us, "us_all_classes": total(ALL)[0] * to_us}


def patch_floor(path, flt):
    (loops, outside), _ = kernel(path, flt)
    step = max((lp for lp in loops.values() if lp["mix"].get("mfma16")), key=lambda lp: lp["n"])
    clk = issue(step["mix"])
    return {"step_clk": clk, "all_clk": issue(step["mix"], ALL), "us": {m: clk * n / SIMDS / GHZ / 1e3 for m, n in STEPS.items()}}


def main():
    prod, abl = assembly([]), assembly(["-DMSDA_ABLATION"])
    rows = []
    for label, flt in (("records backward <2, swap>", "cell_records_backward_kernel<2, true>"),
                       ("records backward <2, no swap>", "cell_records_backward_kernel<2, false>"),
                       ("cell forward <2, 0>", "cell_forward_kernel<2, 0>"),
                       ("cell forward <2, 3> (EMIT)", "cell_forward_kernel<2, 3>")):
        f = cell_kernel_floor(prod, flt)
        rows.append((label, f))
        print(f"{label:32s} level-iteration {f['level_clk']:6.0f} clk, staging round {f['staging_round_clk']:4.0f}, outside {f['outside_clk']:5.0f}"
              f" -> {f['per_wave_clk']:6.0f} VALU clk per wave = {f['us']:5.1f} us per N = 4 call (matrix pipe {f['mfma_us']:4.1f} us; "
              f"counting every instruction class: {f['us_all_classes']:5.1f} us)")
    for label, path, flt in (("patch pass, product", prod, "patch_dest_kernel<unsigned short, 4>"),
                             ("patch pass, MULTI arm", abl, "patch_dest_multi_kernel<unsigned short, 4, false>"),
                             ("patch pass, MULTI + CELLG arm", abl, "patch_dest_multi_kernel<unsigned short, 4, true>")):
        f = patch_floor(path, flt)
        rows.append((label, f))
        print(f"{label:32s} step {f['step_clk']:5.0f} VALU issue clk (every class: {f['all_clk']:4.0f}) -> "
              + ", ".join(f"{f['us'][m]:5.1f} us ({m} locations)" for m in STEPS) + "  [measured: 256 / 264 us]")
    rb, pp = rows[0][1]["us"], rows[-1][1]["us"]["model"]
    print(f"whole encoder backward on the proposed route (records backward + CELLG patch pass, model-like locations): "
          f"{rb + pp:.0f} us at full issue = {409.6e6 / ((rb + pp) * 1e-6) / 8e12:.2f} of 8 TB/s; at the 2.3-2.9 x of today's kernels "
          f"{2.3 * (rb + pp):.0f}-{2.9 * (rb + pp):.0f} us = {409.6e6 / (2.9 * (rb + pp) * 1e-6) / 8e12:.2f}-{409.6e6 / (2.3 * (rb + pp) * 1e-6) / 8e12:.2f}"
          f"   [today, measured: 542 us = 0.094]")
    return rows


if __name__ == "__main__":
    main()
