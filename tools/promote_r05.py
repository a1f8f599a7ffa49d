"""Reads the `experiments` object of a bench.py line (or the JSON tools/experiments_r05.py prints) and says, arm by arm, what the
numbers decide -- so that the session that finally has a GPU spends its minutes on the edits, not on reading tables.

    python tools/promote_r05.py gpurun_out/r06/experiments.json         # tools/experiments_r05.py [--all | --arms | --uniform-arms]
    python tools/promote_r05.py gpurun_out/experiments_last.json        # what `bench.py --experiments` leaves (or its stderr log:
                                                                        # the `EXPERIMENTS {json}` line is found too)

Rules (the ones DESIGN.md section 8 states): an arm of the MSDA backward is promoted only if its grad_value is BIT-EQUAL to the
product kernels' and its other gradients agree within a rounding of their type (`accepted`; bit-equal digests imply it) and it is
faster by more than the noise margin (3 %); kernels that replace PyTorch op sequences need
agreement within their test tolerance and a speed-up.  Prints one line per candidate: verdict, measured numbers, and the single
place to edit.  Exit code 0 always (a report, not a gate)."""
import json
import sys

MARGIN = 0.03


def load(path):
    last = None
    for line in open(path):
        line = line.strip()
        if line.startswith("EXPERIMENTS {"):       # bench.py --experiments writes the object to stderr behind this word
            line = line[len("EXPERIMENTS "):]
        if line.startswith("{"):
            try:
                last = json.loads(line)
            except ValueError:
                continue
    if last is None:
        raise SystemExit(f"{path}: no JSON object found")
    rep = dict(last.get("experiments", last))
    if "experiments" in last:                      # a bench line: keep what the parent measured itself next to its experiments
        rep["_parent"] = {"ms_per_step": last.get("ms_per_step"), "roofline_frac": last.get("roofline", {}).get("frac"),
                          "mean_launch_us": last.get("roofline", {}).get("mean_launch_us")}
    return rep


def faster(new, old):
    return new is not None and old is not None and new < old * (1.0 - MARGIN)


def decide(rep):
    out = []
    arms = rep.get("encoder_backward_arms", {})
    base = next(iter(arms.values()), None) if arms else None
    if isinstance(base, dict) and "fused" in base:
        b_us = base["fused"]["us"]
        for name, v in list(arms.items())[1:]:
            if "error" in v:
                out.append(("SKIP", name, v["error"], ""))
                continue
            # (accepted: grad_value bit-equal + the formula gradients within a rounding, tools/experiments_r05.py; older reports: digests)
            ok = all(v[c].get("accepted", v[c].get("equal_bits")) for c in ("fused", "b0")) and v["fused"].get("finite")
            us = v["fused"]["us"]
            where = ("csrc/msda_patch.hip: the arm's kernel / launch out of `#ifdef MSDA_ABLATION` (patch_dest_multi_kernel<., ., true> + "
                     "grad_out_cells_kernel, gcell_bytes), rebuild, GPU suite" if "cellg" in name else
                     "csrc/msda_patch.hip: kCellMode / kPatchMulti, rebuild, GPU suite")
            if not ok:
                out.append(("REJECT", name, f"gradients differ from the product kernels' ({us} us)", "delete the arm"))
            elif faster(us, b_us):
                out.append(("PROMOTE", name, f"{b_us} -> {us} us per fused backward, " + ("bit-equal" if v["fused"].get("equal_bits") else
                            "grad_value bit-equal, the formula gradients within a rounding"), where))
            else:
                out.append(("KEEP OFF", name, f"{b_us} -> {us} us: not faster", "delete the arm"))
    uni = rep.get("uniform_location_arms", {})
    ubase = uni.get("default", {})
    for name, v in list(uni.items())[1:]:
        if "error" in v or "error" in ubase:
            out.append(("SKIP", f"uniform locations: {name}", v.get("error") or ubase.get("error"), ""))
            continue
        u_old, u_new, m_old, m_new = ubase["uniform"]["us"], v["uniform"]["us"], ubase["model"]["us"], v["model"]["us"]
        detail = f"uniform locations {u_old} -> {u_new} us per backward (round 2: 975-1 130), model-like {m_old} -> {m_new} us"
        if not (v["uniform"].get("accepted") and v["model"].get("accepted")):
            out.append(("REJECT", f"uniform locations: {name}", "gradients differ or are not repeatable: " + detail, "delete the arm"))
        elif faster(u_new, u_old) and m_new <= m_old * (1.0 + MARGIN):
            out.append(("PROMOTE", f"uniform locations: {name}", detail,
                        "csrc/msda_patch.hip (the far-return block), csrc/msda_quad.hip (launch_quad_backward_gated), csrc/msda_dest.hip "
                        "(bin_queue_kernel / combine_queue_kernel + their launches): out of `#ifdef MSDA_ABLATION`, rebuild, GPU suite"))
        else:
            out.append(("KEEP OFF", f"uniform locations: {name}", detail, "delete the arm"))
    rc = rep.get("encoder_records_route_cellg", {})
    if rc and "error" not in rc and "records_swap" in rc and "product" in rc:
        v, pr = rc["records_swap"], rc["product"]
        ok = v.get("accepted") and v.get("finite")
        out.append(("INFO" if ok else "REJECT", "records route + CELLG patch pass (fused cell-major copy)",
                    f"backward {pr['bwd_us']} (product cell kernel + copy kernel + CELLG) -> {v['bwd_us']} us, forward {v['fwd_us']} us; "
                    "compare with the plain records route above", "both are never-run code: promote the pair or neither"))
    rec = rep.get("encoder_records_route", {})
    if "error" in rec:
        out.append(("SKIP", "records route", rec["error"], ""))
    elif rec:
        p = rec["product"]
        for name in ("records", "records_swap"):
            v = rec.get(name)
            if not v:
                continue
            pair_old, pair_new = p["fwd_us"] + p["bwd_us"], v["fwd_us"] + v["bwd_us"]
            detail = (f"forward {p['fwd_us']} -> {v['fwd_us']} us, backward {p['bwd_us']} -> {v['bwd_us']} us, pair {pair_old:.0f} -> "
                      f"{pair_new:.0f} us; far flag {v.get('far_flag')}, {v.get('records_MB')} MB of records")
            if not (v.get("accepted", v.get("equal_bits")) and v.get("finite")):
                out.append(("REJECT", f"records route ({name})", "gradients differ from the product kernels': " + detail,
                            "tests/test_zzz_records_gpu.py names the tensor"))
            elif faster(pair_new, pair_old):
                out.append(("PROMOTE", f"records route ({name})", detail,
                            "rlipv2_amd/msda.py: records_route = True, records_swap = " + str(name == "records_swap")
                            + "; then bench.py --set msda.records_route=1 for the step, GPU suite"))
            else:
                out.append(("KEEP OFF", f"records route ({name})", detail, ""))
        st = rep.get("train_step_with_records_route")
        if st and "error" not in st:
            par = rep.get("_parent", {})
            out.append(("PROMOTE" if faster(st.get("ms_per_step"), par.get("ms_per_step")) else "INFO", "train step with the records route",
                        f"{par.get('ms_per_step')} -> {st.get('ms_per_step')} ms per step, roofline.frac {par.get('roofline_frac')} -> "
                        f"{st.get('roofline_frac')} ({st.get('roofline_kernel')}: {par.get('mean_launch_us')} -> {st.get('mean_launch_us')} us)",
                        "the step-level number that decides; same edit as above"))
        elif st:
            out.append(("SKIP", "train step with the records route", st["error"], ""))
        c = rec.get("cell_forward")
        if c:
            out.append(("INFO", "cell forward alone", f"forward {p['fwd_us']} -> {c['fwd_us']} us, output differs by "
                        f"{c.get('out_max_diff_rel_to_max'):.2e} of the maximum", "msda.fused_forward_cell (bench.py --msda-fwd-cell)"))
    fwd = rep.get("encoder_forward_cell", {})
    for mode, v in fwd.items() if "error" not in fwd else []:
        if isinstance(v, dict) and "cell_us" in v:
            verdict = "PROMOTE" if faster(v["cell_us"], v["quad_us"]) and v["max_diff_rel_to_max"] <= 2.0 ** -6 and not v["non_finite"] else "KEEP OFF"
            out.append((verdict, f"cell forward, B0 signature ({mode} locations)", f"{v['quad_us']} -> {v['cell_us']} us, max diff "
                        f"{v['max_diff_rel_to_max']:.2e} of the maximum", "csrc/msda_api.hip: the AUTO choice of msda_forward_hs for bfloat16 encoder calls"))
    stp = rep.get("decoder_cross_attention_sample_then_project", {})
    if "standard_us" in stp:
        good = max(stp["rel_l2_out"], stp["rel_l2_d_src"], stp["rel_l2_d_value_proj_weight"]) <= 3e-2
        verdict = "PROMOTE" if good and faster(stp["sample_then_project_us"], stp["standard_us"]) else ("REJECT" if not good else "KEEP OFF")
        out.append((verdict, "decoder cross-attention: sample, then project", f"{stp['standard_us']} -> {stp['sample_then_project_us']} us per layer call "
                    f"(forward + backward), rel. L2 of d_src {stp['rel_l2_d_src']:.1e}", "rlipv2_amd/deform_attn.py: sample_then_project = True"))
    elif "error" in stp:
        out.append(("SKIP", "sample, then project", stp["error"], ""))
    sw = rep.get("swin_routes", {})
    st = sw.get("stage0_two_blocks_fwd_bwd")
    if st:
        good = st["finite"] and st["rel_l2_out"] <= 3e-2 and st["rel_l2_dx"] <= 5e-2
        verdict = "PROMOTE" if good and faster(st["fused_us"], st["ops_us"]) else ("REJECT" if not good else "KEEP OFF")
        out.append((verdict, "Swin stage 0: fused window attention + wide LayerNorm", f"{st['ops_us']} -> {st['fused_us']} us per two blocks (fwd + bwd)",
                    "nothing to edit: routes.validate switches both on per run (config.host_routes); delete the kernels if REJECT"))
        for name, v in sw.get("stage0_weight_gradients", {}).items():
            verdict = "PROMOTE" if v["rel_l2"] <= 2e-2 and faster(v["padded_kernel_us"], v["library_us"]) else "KEEP OFF"
            out.append((verdict, f"padded weight gradient {name}", f"{v['library_us']} -> {v['padded_kernel_us']} us", "rlipv2_amd/linear.py: pad_wgrad_to_128 = True"))
    elif "error" in sw:
        out.append(("SKIP", "Swin routes", sw["error"], ""))
    return out


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    rows = decide(load(sys.argv[1]))
    if not rows:
        print("no experiments in this file (bench.py ran with --no-experiments, under a profiler, or on more than one GPU)")
    for verdict, name, detail, where in rows:
        print(f"{verdict:9s} {name}: {detail}" + (f"\n          -> {where}" if where else ""))


if __name__ == "__main__":
    main()
