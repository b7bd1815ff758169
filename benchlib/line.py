"""The ONE stdout line of bench.py.

The driver's `parsed` record (BENCH_rNN.json) keeps the contract keys; of the dicts among them (`config`, `roofline`,
`cpu_baseline`) ONE level of scalars, strings cut at 120 characters, AT MOST 24 KEYS PER DICT (BENCH_r05.json: 26 were
emitted, `ms_layernorm` / `ms_index` were dropped), nested dicts / lists dropped; every other top-level key survives only
by name.  So everything a reader of that record needs is a flat scalar under one of the three dicts, and each of them
stays at or below `DICT_KEY_BUDGET` keys (tests/test_bench_contract.py models the cap)."""
import json

COMPACT_LIMIT = 6000          # bytes; the driver keeps the last 8 KB of stdout (a real N = 1 line is ~3.5 KB)
DRIVER_DICT_CAP = 24          # keys the driver's record keeps of `config` / `roofline` / `cpu_baseline`
DICT_KEY_BUDGET = 22          # what this file allows itself (two keys of slack under the cap)

ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch")
DOMINANT_MAP = (("name", "dominant_kernel"), ("frac", "dominant_frac"), ("tflops", "dominant_tflops"),
                ("mean_us", "dominant_mean_us"), ("launches_per_step", "dominant_launches_per_step"),
                ("gflop_per_step", "dominant_gflop_per_step"))
CLASSES_MS = ("gemm", "union_conv", "mask_conv", "attention", "layernorm", "index")
MS_NOTE = "ms_gemm = nn.Linear + conv3x3 + union conv (one fused launch); ms_union_conv = its leftover-tile launch only"
assert len(MS_NOTE) <= 118


def error_line(args, world, msg):
    """A compact line for a run that could not start (no device for a rank): every contract key, value 0, and `error`."""
    return json.dumps({"metric": "frames/sec (PredCls inference)", "value": 0.0, "unit": "frames/s", "n_gpus": world,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "not run"},
                       "roofline": None, "cpu_baseline": None, "error": msg[:200]})


def _r(x, nd=4):
    """floats rounded for the compact line (the detail file keeps full precision)"""
    if isinstance(x, float):
        return round(x, nd) if abs(x) < 1e6 else round(x, 1)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(d):
    """The ONE stdout line: every contract field + scalars of the extra legs, no tables, no prose; < COMPACT_LIMIT bytes.
    `value`, `ms_per_step` and the roofline numbers keep full precision (the driver and the contract tests recompute
    them); everything else is rounded."""
    cfg = d["config"]
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                             "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": f"synthetic {cfg['frames_per_clip']}x{cfg['boxes_per_frame']}x2048-d clips, "
                                 + ("STTran PredCls" if "PredCls" in d["metric"] else "DSG-DETR sgdet") + " forward, inputs in HBM",
                     "clips_per_step": cfg["clips_per_step"], "frames_per_clip": cfg["frames_per_clip"],
                     "boxes_per_frame": cfg["boxes_per_frame"], "pairs_per_step": cfg["pairs_per_step"],
                     "batch": "per-clip pointer tables, formed inside every timed step" if cfg["clips_per_step"] > 1 else "one clip",
                     "layout_cache": cfg["layout_cache"].split(":")[0].split(" (")[0], "hip_graph": cfg["hip_graph"],
                     "sharding": f"whole clips over {d['n_gpus']} rank(s), one all-gather of [pairs,26] rows per step"
                                 if d["n_gpus"] > 1 else "single GPU"}
    # the scaling scalars (main(): `scaling_scalars`) live INSIDE config, flat: that is where the driver's record keeps them
    for k, v in (d.get("scaling_scalars") or {}).items():
        out["config"][k] = _r(v, 4) if isinstance(v, float) and abs(v) < 100 else _r(v, 1)
    out["repeats"] = _r(d["repeats"], 1)
    out["ranks_seen"], out["distinct_devices"] = d["ranks_seen"], d["distinct_devices"]
    if "roofline" in d:
        r = d["roofline"]
        out["roofline"] = {k: r[k] for k in ROOFLINE_KEYS}
        # FLAT scalars only: the driver's parsed record keeps one level of scalars under `roofline` / `cpu_baseline` and
        # drops nested dicts (BENCH_r04.json lost `dominant{}`) and every key past the 24th (BENCH_r05.json lost
        # `ms_layernorm` / `ms_index`), so the dominant kernel's row and the per-class times of one step are spelled out as
        # `dominant_*` / `ms_*` keys -- 20 keys in all; launches_per_step, avg_launch_us, share_of_device_time and the
        # nested tables live in the detail file
        dom = r.get("dominant") or {}
        for src, dst in DOMINANT_MAP:
            out["roofline"][dst] = dom.get(src)
        for cls in CLASSES_MS:
            out["roofline"]["ms_" + cls] = _r(float(r["per_class_ms_per_step"].get(cls, 0.0)), 4)
        # (since round 6 both convolutions of the pair fusion run as ONE launch in the gemm class: a reader of the record must
        #  not take `ms_union_conv` = 0 for skipped work -- said in `config`, which has the room: `roofline` stays at 20 keys)
        out["config"]["kernel_classes"] = MS_NOTE
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "host_cores": c["host_cores"],
                               "kind": c["kind"], "sample": c["sample"][:118]}
        for k in ("impl", "numpy_value", "numpy_cores", "torch_value", "torch_cores", "torch_value_8_threads",
                  "torch_value_64_threads", "torch_value_all_threads"):
            if k in c:
                out["cpu_baseline"][k] = _r(c[k], 2)
    for k in ("one_clip_per_pass", "one_clip_coalesced", "same_batch", "two_steps_in_flight", "pcie_inclusive_overlapped",
              "one_rank_alone"):
        if k in d:
            out[k] = {"value": _r(d[k]["value"], 1), "ms_per_step": _r(d[k]["ms_per_step"], 4)}
    if "one_clip_per_pass" in d and "serial" in d["one_clip_per_pass"]:
        o = d["one_clip_per_pass"]
        out["one_clip_per_pass"]["lanes"] = o["lanes"]
        out["one_clip_per_pass"]["serial"] = _r(o["serial"]["value"], 1)
    if "one_clip_coalesced" in d:
        o = d["one_clip_coalesced"]
        out["one_clip_coalesced"].update(coalesce=o["coalesce"], lanes=o["lanes"], no_hints=_r(o["no_hints"]["value"], 1),
                                         result_latency_ms=_r(o.get("result_latency_ms"), 3))
    if "pcie_inclusive_overlapped" in d:
        out["pcie_inclusive_overlapped"]["h2d_gb_per_s"] = _r(d["pcie_inclusive_overlapped"]["h2d_gb_per_s"], 1)
    if "allgather_ms" in d:
        out["allgather_ms"], out["allgather_bytes_per_rank"] = _r(d["allgather_ms"]), d["allgather_bytes_per_rank"]
    if "rccl_selftest" in d:
        st = d["rccl_selftest"]
        out["rccl_selftest"] = {k: _r(st[k]) for k in ("ok", "backend", "allgather_ms", "gather_verified", "seconds") if k in st}
        if "error" in st:
            out["rccl_selftest"]["error"] = str(st["error"])[:120]
    if "batch_sweep" in d:
        out["batch_sweep"] = {str(b["clips_per_step"]): _r(b["value"], 1) for b in d["batch_sweep"]}
    if "reference_arithmetic" in d:
        out["reference_arithmetic_frac"] = _r(d["reference_arithmetic"]["frac_of_fp32_mfma_peak"])
    w = {}
    for name, blk in d.get("workloads", {}).items():
        if "error" in blk:
            w[name] = {"error": blk["error"][:120]}
            continue
        e = {"value": _r(blk["value"], 1)}
        if "ms_per_step" in blk:
            e["ms_per_step"] = _r(blk["ms_per_step"], 3)
        if "roofline" in blk:
            e["roofline_frac"] = _r(blk["roofline"]["frac"])
        if "cpu_baseline" in blk:
            e["cpu_baseline"] = _r(blk["cpu_baseline"]["value"], 1)
        if "two_steps_in_flight" in blk:
            e["two_steps_in_flight"] = _r(blk["two_steps_in_flight"]["value"], 1)
        if "one_clip_per_pass" in blk:
            e["one_clip_per_pass"] = _r(blk["one_clip_per_pass"]["value"], 1)
            if "serial" in blk["one_clip_per_pass"]:
                e["one_clip_serial"] = _r(blk["one_clip_per_pass"]["serial"]["value"], 1)
        if "one_clip_coalesced" in blk:
            e["one_clip_coalesced"] = _r(blk["one_clip_coalesced"]["value"], 1)
        if "max_abs_diff_vs_fp32_engine" in blk:
            e["max_abs_diff_vs_fp32_engine"] = blk["max_abs_diff_vs_fp32_engine"]
        if "allgather_ms" in blk:
            e["allgather_ms"] = _r(blk["allgather_ms"])
        if "one_rank_alone" in blk:
            e["one_rank_alone"] = _r(blk["one_rank_alone"]["value"], 1)
        w[name] = e
    if w:
        out["workloads"] = w
    ss = {}
    for name, blk in d.get("strong_scaling", {}).items():
        if "error" in blk:
            ss[name] = {"error": blk["error"][:120]}
            continue
        ss[name] = {"value": _r(blk["value"], 1), "seconds": _r(blk["seconds"]), "clips": blk["clips"], "frames": blk["frames"],
                    "ranks": blk["ranks"], "busy_max_s": _r(blk["busy_max_s"]), "eval_max_s": _r(blk["eval_max_s"]),
                    "eval_s_rank0": _r(blk["eval_s_rank0"]), "lpt_imbalance": _r(blk["lpt_imbalance"]),
                    "busy_imbalance": _r(blk["busy_imbalance"]), "gather_verified": blk["gather_verified"],
                    "R@20": blk["recall_with_constraint"].get("20")}
        if "rank0_alone" in blk:
            ss[name]["rank0_alone"] = _r(blk["rank0_alone"]["value"], 1)
    if ss:
        out["strong_scaling"] = ss
    out["detail"] = "bench_detail.json + stderr BENCH_DETAIL: per-kernel / per-shape tables"
    for blk in ("config", "roofline", "cpu_baseline"):
        if isinstance(out.get(blk), dict) and len(out[blk]) > DICT_KEY_BUDGET:
            raise RuntimeError(f"compact bench line: {len(out[blk])} keys under `{blk}` (the driver's record keeps {DRIVER_DICT_CAP})")
    line = json.dumps(out)
    for k in ("batch_sweep", "reference_arithmetic_frac", "same_batch", "two_steps_in_flight", "detail"):   # never expected; keeps the promise
        if len(line) < COMPACT_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out)
    if len(line) >= COMPACT_LIMIT:
        raise RuntimeError(f"compact bench line is {len(line)} bytes (limit {COMPACT_LIMIT})")
    return line
