"""Numeric model of the LDS bank conflicts of the kernels' transposing reads under the per-instruction rules of
/opt/skills/guides/MI355X_MICROARCH.md (section LDS): `ds_read_b64_tr_b16` is serviced in two 32-lane groups, the bank of byte
address a is (a / 4) mod 64, identical addresses broadcast, N distinct addresses on one bank within a group cost N cycles.
Prints, per layout, the conflict degree (cycles per lane group; 1 = conflict-free) of

  * patch_dest_kernel's 12 transposing reads per 32-group step (csrc/msda_patch.hip): product layout vs the experimental
    instantiation (odd 8-row blocks staged with their halves exchanged);
  * cell_forward_kernel's corner-row reads (csrc/msda_cell_forward.inc): both lane groups of a half on the same channel half
    (first version) vs the odd group on the other half (as built), over random windows and sample positions.

The guide warns that the transposing read has further conflict classes; the counters (SQ_LDS_BANK_CONFLICT) have the last word
(tools/gpu_triage_r06.sh collects them).  CPU only.   usage: python tools/lds_bank_model.py
"""
import numpy as np


def degree(addrs):
    """addrs: byte addresses of one lane group's 8-byte reads -> cycles (max distinct dword addresses on one of 64 banks)"""
    banks = {}
    for a in addrs:
        for w in range(2):
            d = (a >> 2) + w
            banks.setdefault(d % 64, set()).add(d)
    return max(len(v) for v in banks.values())


def patch_reads(multi):
    """conflict degree of every (32-lane half, read) of a step of patch_dest_kernel<., ., multi>"""
    k_off_g, k_off_a = 768, 768 + 32 * 64                       # kOffG, kOffA of the per-wave LDS carve-up
    out = []
    for half in range(2):
        for which in [("g", t, j) for t in range(2) for j in range(2)] + [("a", m, j) for m in range(4) for j in range(2)]:
            addrs = []
            for lane in range(half * 32, half * 32 + 32):
                p16, kg = lane & 15, lane >> 4
                rsw = (kg & 1) if multi else 0
                a_j, g_t = (-128 if rsw else 128), (-32 if rsw else 32)
                a_rd = k_off_a + (8 * kg + (p16 >> 2) + 4 * rsw) * 32 + (p16 & 3) * 8
                g_rd = k_off_g + (8 * kg + (p16 >> 2)) * 64 + (p16 & 3) * 8 + rsw * 32
                kind, x, j = which
                addrs.append(g_rd + x * g_t + j * 256 if kind == "g" else a_rd + x * 1024 + j * a_j)
            out.append(degree(addrs))
    return out


def forward_reads(swap, trials=5000, seed=0):
    """mean conflict degree of a 32-lane half of cell_forward_kernel's corner-row read: two queries x four corner rows"""
    rng = np.random.default_rng(seed)
    tot = 0
    for _ in range(trials):
        pitch = int(rng.integers(3, 30))
        pitch += (2 - pitch) & 3                                  # = 2 (mod 4), as the kernel pads it
        rows = int(rng.integers(3, 30))
        addrs = []
        for g in range(2):
            y, x = int(rng.integers(0, rows - 1)), int(rng.integers(0, pitch - 1))
            base = (y * pitch + x) * 64
            for p in range(16):
                crn, r4 = p >> 2, p & 3
                addrs.append(base + ((crn >> 1) * pitch + (crn & 1)) * 64 + r4 * 8 + ((g & 1) * 32 if swap else 0))
        tot += degree(addrs)
    return tot / trials


def main():
    print("patch_dest_kernel, 24 (half, read) pairs of a step:")
    print("   product layout            ", patch_reads(False))
    print("   experimental instantiation", patch_reads(True))
    print("cell_forward_kernel, corner-row read of a 32-lane half (mean over random windows / samples):")
    print(f"   both lane groups on the same channel half  {forward_reads(False):.2f}")
    print(f"   odd lane group on the other half (built)   {forward_reads(True):.2f}")


if __name__ == "__main__":
    main()
