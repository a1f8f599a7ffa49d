"""Static bars of the records route's "record diet"  -- what can be pinned without a GPU:
the saved state the forward leaves per N = 4 encoder call (<= 230 MB; the 16-byte records of round 5 made it 408 MB) and the
instruction count of the backward's sample loop read off the gfx950 assembly (<= 26 VALU instructions per sample and lane).
Bit-equality with the product kernels on the lane-level model is tests/test_records_emulated.py."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIPCC = "/opt/rocm/bin/hipcc"
PYRAMID = [(100, 167), (50, 84), (25, 42), (13, 21)]          # 800 x 1333


def test_saved_state_of_an_encoder_call_fits_the_bar():
    """msda_records_bytes is host arithmetic of the product library (no GPU call): N = 4, 8 heads, the bench pyramid"""
    from rlipv2_amd import _lib
    L = _lib.lib()
    hs = (ctypes.c_int64 * 8)(*[v for hw in PYRAMID for v in hw])
    S = sum(h * w for h, w in PYRAMID)
    total = int(L.msda_records_bytes(_lib.MSDA_BF16, hs, 4, S, 8, 32, 4, S, 4))
    assert 0 < total <= 230e6, total
    cells = 7 * 11
    items = 4 * 8 * cells
    sample_records = items * 4 * 352 * 4 * 2                   # [item][level][352 query slots][point] x 2 bytes
    group_records = items * 4 * 340 * 48                       # what patch_dest_kernel reads anyway
    masks = total - 256 - items * 128 - sample_records - group_records
    assert sample_records < 30e6 and 30e6 < masks < 40e6, (sample_records, masks)
    # against the float32 locations / weights the product route saves for the same call (136.5 MB) + the 194 MB of masks and group
    # records its backward writes and reads back: the pair moves less saved state than the product route
    assert total < 136.5e6 + group_records + masks


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_sample_loop_instruction_count_from_the_device_assembly(tmp_path):
    from tools import isa_blocks
    asm = str(tmp_path / "msda_patch.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-w",
                    os.path.join(ROOT, "rlipv2_amd", "csrc", "msda_patch.hip"), "-o", asm], check=True, timeout=600)
    for swap, bar in (("true", 26.0), ("false", 30.0)):       # (without the operand swap: 4 DPP broadcasts per sample on top)
        (name, blocks), = isa_blocks.blocks_of(asm, f"cell_records_backward_kernel<2, {swap}>")
        staged = [c for _, c, ins in blocks if c.get("mfma4") == 32 and sum(t.startswith("ds_read_b128") for t in ins) == 16]
        assert len(staged) == 3, [dict(c) for c in staged]     # one block per group of 16 queries
        # a block = 16 queries x 4 samples of one level: every lane is the corner of 4 samples -> per sample and lane.  The first
        # block also holds the level's set-up (record loads of all groups, staging tail); the steady state is the other two
        per_sample = sorted((c["valu"] + c["valu_slow"]) / 4.0 for c in staged)
        assert per_sample[1] <= bar, (swap, per_sample)
        assert all(c.get("scratch", 0) == 0 for c in staged[1:]), "a spill inside the sample loop"
