"""tools/lds_bank_model.py: the LDS bank arithmetic DESIGN.md quotes for the transposing reads (per-instruction lane groups and
bank function of MI355X_MICROARCH.md) stays reproducible -- and stays in step with the kernels' address constants."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lds_bank_model as B  # noqa: E402


def test_patch_pass_reads_are_two_way_in_the_product_layout_and_conflict_free_with_exchanged_halves():
    assert B.patch_reads(False) == [2] * 24
    assert B.patch_reads(True) == [1] * 24


def test_forward_corner_reads_are_conflict_free_with_the_odd_lane_group_on_the_other_channel_half():
    assert B.forward_reads(True, trials=500) == 1.0
    assert B.forward_reads(False, trials=500) > 1.9


def test_model_uses_the_kernel_s_lds_carve_up():
    src = open(os.path.join(ROOT, "rlipv2_amd", "csrc", "msda_patch.hip")).read()
    cap = int(re.search(r"constexpr int kListCap = (\d+);", src).group(1))
    assert "kOffG = kListCap * 2, kOffA = kOffG + kStep * 64" in src and cap * 2 == 768      # the offsets patch_reads() assumes
    assert (768 + 32 * 64 + 4 * 32 * 32) % 256 == 0                                           # per-wave blocks keep the bank phase
