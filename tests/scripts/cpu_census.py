"""Operator census of the host-side glue on the CPU: the small ParSeDA of the CPU tests (6-dec-layer option) + the set
criterion, forward and backward under the autograd profiler.  Every aten operator that computes (views excluded) is one
kernel launch on the GPU, so this counts -- without a GPU -- what the decoders, heads and criterion add to the launch tail
of the train step (the fused HIP functions are replaced by their torch twins here and are not what this is about).
usage: python tests/scripts/cpu_census.py [dec_layers]   (lives under tests/: it runs the modules with the oracle-backed op)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_modules_cpu as T  # noqa: E402

from rlipv2_amd import criterion as MC  # noqa: E402
from rlipv2_amd import deform_attn, parseda  # noqa: E402

VIEWS = {"aten::view", "aten::reshape", "aten::permute", "aten::transpose", "aten::t", "aten::expand", "aten::unsqueeze",
         "aten::squeeze", "aten::select", "aten::slice", "aten::as_strided", "aten::detach", "aten::alias", "aten::unbind",
         "aten::split", "aten::split_with_sizes", "aten::chunk", "aten::narrow", "aten::_unsafe_view", "aten::flatten",
         "aten::unflatten", "aten::expand_as", "aten::view_as", "aten::empty", "aten::empty_like", "aten::empty_strided",
         "aten::result_type", "aten::to", "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense", "aten::size",
         "aten::is_nonzero", "aten::stride", "aten::numpy_T", "aten::contiguous", "aten::movedim", "aten::swapaxes",
         "aten::resolve_conj", "aten::resolve_neg", "aten::broadcast_tensors", "aten::set_", "aten::unsafe_split",
         "aten::unsafe_chunk", "aten::type_as", "aten::new_empty", "aten::new_zeros", "aten::new_ones", "aten::new_full",
         "aten::zeros_like", "aten::ones_like", "aten::full_like", "aten::zeros", "aten::ones", "aten::full",
         "aten::linear", "aten::matmul", "aten::einsum", "aten::layer_norm", "aten::softmax", "aten::log_softmax",
         "aten::dropout", "aten::feature_dropout", "aten::cross_entropy_loss", "aten::nll_loss_nd", "aten::nll_loss",
         "aten::scaled_dot_product_attention", "aten::group_norm", "aten::batch_norm", "aten::conv2d", "aten::convolution",
         "aten::relu_", "aten::l1_loss", "aten::pairwise_distance", "aten::cdist", "aten::max_pool2d", "aten::clamp_min_",
         "aten::where", "aten::sum_to_size", "aten::_to_copy"}      # composite wrappers: their leaves are counted instead


def main():
    dec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    g = T.load("parseda")
    args = parseda.default_args(num_queries=20, enc_layers=2 * dec, dec_layers=dec, dim_feedforward=512, pseudo_verb=True)      # (one language state per fusion = per decoder layer, as the scripts have it)
    bb = T._FeatureBackbone((32, 64, 128))
    model = parseda.build_parseda(bb, args).eval()
    T.fill_closed_form(model)
    from make_model_golden import criterion_case
    _, _, targets, _ = criterion_case()
    crit = MC.SetCriterionHOI(MC.HungarianMatcherHOI(1, 1, 2.5, 1, subject_class=True), MC.build_weight_dict(dec))
    with torch.autograd.profiler.profile() as prof:
        mc, out, feats, _ = T.run_small_parseda(model, bb, g)
        n_cls = out["pred_obj_logits"].shape[-1]
        tg = []
        for t in targets:                                                   # labels of the criterion case, clipped to this model
            t = dict(t)
            for k in ("obj_labels", "sub_labels"):
                t[k] = t[k].clamp(max=n_cls - 2)
            t["verb_labels"] = t["verb_labels"][:, :out["pred_verb_logits"].shape[-1] - 0][:, :out["pred_verb_logits"].shape[-1]]
            tg.append(t)
        try:
            ld = crit(out, tg)
            total = crit.weighted_sum(ld)
        except Exception as e:                                             # noqa: BLE001 -- shapes of the stored case may not fit
            print("criterion skipped:", repr(e)[:200])
            total = sum((out[k] ** 2).sum() for k in T.KEYS)
        total.backward()
    fwd, bwd_nodes = collections.Counter(), collections.Counter()
    by_node = collections.defaultdict(collections.Counter)
    # leaves only: an aten op with no aten children
    events = [e for e in prof.function_events]
    for e in events:
        if not e.name.startswith("aten::"):
            continue
        if any(c.name.startswith("aten::") and c.name not in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::as_strided", "aten::view", "aten::expand", "aten::resize_", "aten::result_type", "aten::to", "aten::_to_copy", "aten::select", "aten::slice", "aten::reshape", "aten::transpose", "aten::permute", "aten::unsqueeze", "aten::squeeze", "aten::_unsafe_view", "aten::t", "aten::stride", "aten::detach", "aten::alias")
               for c in e.cpu_children):
            continue
        if e.name in VIEWS:
            continue
        p = e.cpu_parent
        node = None
        while p is not None:
            if "Backward" in p.name or p.name.startswith("autograd::engine"):
                node = p.name
                if "Backward" in p.name:
                    break
            p = p.cpu_parent
        if node and "Backward" in node:
            bwd_nodes[node.split(": ")[-1]] += 1
            by_node[node.split(": ")[-1]][e.name] += 1
        else:
            fwd[e.name] += 1
    print(f"decoder layers {dec}: forward compute ops {sum(fwd.values())}, backward compute ops {sum(bwd_nodes.values())}")
    print("forward ops:")
    for k, v in fwd.most_common(40):
        print(f"   {v:5d}  {k}")
    print("backward nodes (compute ops inside):")
    for k, v in bwd_nodes.most_common(40):
        inner = ", ".join(f"{a.replace('aten::', '')} x{b}" for a, b in by_node[k].most_common(4))
        print(f"   {v:5d}  {k:45s} {inner}")


if __name__ == "__main__":
    main()
