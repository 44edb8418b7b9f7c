"""The ONE JSON line `bench.py` prints, kept small enough for any log-tail reader.

`bench.py` measures many legs (batch sweep, product modes, per-step parity tables, stage times ...).  All of that goes to
`bench_detail.json` (written beside the repo's `gpurun_out/`, and copied to `profiles/rNN_bench_detail.json` by hand); the line on stdout
carries only what SURVEY.md 8(d) names: metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / config, the `roofline` of
the dominant kernel, the `cpu_baseline`, `parity`, one five-number object per secondary leg and a flat `summary`.

Limits asserted here and in `tests/test_host_cpu.py::test_bench_line_is_compact`: the serialised line is below `MAX_LINE_BYTES`, no string
is longer than `MAX_STRING_CHARS`, and it survives a `json.loads` round trip.
"""
import json
import math

MAX_LINE_BYTES = 6144
MAX_STRING_CHARS = 160


def _num(v, digits=6):
    """Floats to `digits` significant digits (the full precision lives in bench_detail.json); anything non-finite becomes None."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        return float(f"{v:.{digits}g}")
    return v


def _get(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d or d[p] is None:
            return default
        d = d[p]
    return d


def _clip(s, n=MAX_STRING_CHARS):
    if not isinstance(s, str):
        return s
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    """{k: d[k]} for the scalar keys present; strings clipped, floats rounded."""
    out = {}
    for k in keys:
        v = _get(d, k)
        if v is None or isinstance(v, (dict, list)):
            continue
        out[k] = _clip(v) if isinstance(v, str) else _num(v)
    return out


def _roofline(r, short_kernel=None):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "peak_full_mantissa", "frac_of_full_mantissa_peak"))
    out["kernel"] = _clip(short_kernel or str(r.get("kernel", "")).split(" (")[0], 100)
    out.setdefault("traffic", None)
    launch = r.get("avg_launch_ms", r.get("launch_ms"))
    if launch is not None:
        out["launch_ms"] = _num(launch)
    alg = _get(r, "algorithmic", "tflops")
    if alg is None:
        alg = r.get("fp32_equivalent_tflops", r.get("algorithmic_tflops"))
    if alg is not None:
        out["algorithmic_tflops"] = _num(alg)
    if r.get("launches_per_step") is not None:
        out["launches_per_step"] = r["launches_per_step"]
    return out


def _cpu(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind"))
    out["sample"] = _clip(c.get("sample", ""))
    return out


def _leg(leg, extra=()):
    """A secondary leg reduced to {value, unit, roofline.frac, cpu_baseline.value, parity.psnr_db} (+ a few named scalars)."""
    if not isinstance(leg, dict):
        return None
    out = _pick(leg, ("value", "unit") + tuple(extra))
    frac = _get(leg, "roofline", "frac")
    if frac is not None:
        out["roofline"] = _pick(leg["roofline"], ("bound", "frac", "achieved", "peak", "unit"))
    cpu = _get(leg, "cpu_baseline", "value")
    if cpu is not None:
        out["cpu_baseline"] = _pick(leg["cpu_baseline"], ("value", "unit", "cores", "kind"))
    psnr = _get(leg, "parity", "psnr_db")
    if psnr is not None:
        out["parity"] = _pick(leg["parity"], ("psnr_db", "max_abs"))
    return out


def compact_line(full):
    """`full` = the dict of everything bench.py measured (rounds 1-5 printed it whole); returns the dict that goes to stdout."""
    roof = full.get("roofline") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["value"] = _num(line["value"])
    line["ms_per_step"] = _num(line["ms_per_step"])
    cfg = full.get("config") or {}
    line["config"] = {"workload": _clip(cfg.get("workload", "")), **_pick(cfg, ("global_batch", "parallelism", "gflop_per_sample_step"))}
    line["dtype_note"] = _clip(full.get("dtype_note", ""))
    line["step_tflops"] = _num(full.get("step_tflops"))
    line["roofline"] = _roofline(roof, full.get("roofline_kernel_short"))
    for k in ("fp32_mfma_mode", "fp16_mode", "bf16x3_mode"):
        v = _get(roof, k, "value")
        if v is not None:
            line["roofline"][k + "_value"] = _num(v)
    line["cpu_baseline"] = _cpu(full.get("cpu_baseline"))
    par = full.get("parity")
    if isinstance(par, dict):
        line["parity"] = _pick(par, ("psnr_db", "max_abs", "triplane_psnr_db", "triplane_max_abs"))
        d50 = par.get("ddim50_vs_reference")
        if isinstance(d50, dict):
            line["parity"]["ddim50_max_abs_vs_reference"] = _num(d50.get("max_abs_last", d50.get("max_abs")))
            line["parity"]["ddim50_psnr_db_vs_reference"] = _num(d50.get("psnr_db_last", d50.get("psnr_db")))
        so = par.get("denoise_steps_vs_oracle")
        if isinstance(so, dict):
            line["parity"]["steps_vs_oracle_max_abs"] = _num(so.get("max_abs"))
    else:
        line["parity"] = None
    line["render"] = _leg(full.get("render"), ("ms_per_view", "views_per_gpu"))
    if line["render"] is not None:
        line["render"]["host_inclusive"] = _num(_get(full, "render", "host_inclusive", "value"))
        for k in ("launches_per_view", "hbm_bytes_per_view"):
            v = _get(full, "render", "roofline", k)
            if v is not None:
                line["render"][k] = _num(v)
    line["fit"] = _leg(full.get("fit"), ("ms_per_iteration",))
    line["train"] = _leg(full.get("train"), ("ms_per_step", "batch_per_gpu"))
    e2e = full.get("e2e")
    if isinstance(e2e, dict):
        line["e2e"] = {"value": _num(e2e.get("seconds_per_subject")), "unit": "s/subject", "psnr_db": _num(_get(e2e, "parity", "psnr_db")),
                       "subjects_per_gpu": e2e.get("subjects_per_gpu"),
                       "denoise_steps_per_s": _num(_get(e2e, "sampling", "denoise_steps_per_sec_per_gpu")),
                       "mrays_per_s": _num(_get(e2e, "rendering", "mrays_per_sec_per_gpu"))}
    else:
        line["e2e"] = None
    rccl = full.get("rccl")
    if isinstance(rccl, dict):
        line["rccl"] = _pick(rccl, ("world_size", "backend", "device_count_visible"))
        for k in ("sample_gather", "image_gather_uint8"):
            v = _get(rccl, k, "recv_gb_per_s_per_rank")
            if v is not None:
                line["rccl"][k + "_recv_gb_per_s_per_rank"] = _num(v)
    else:
        line["rccl"] = None
    summ = full.get("summary") or {}
    line["summary"] = {k: (_num(v) if not isinstance(v, str) else _clip(v)) for k, v in summ.items() if not isinstance(v, (dict, list))}
    line["detail"] = full.get("detail_file", "bench_detail.json")
    check_line(line)
    return line


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for k, v in o.items():
            yield k
            yield from _strings(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _strings(v)


def check_line(line):
    s = json.dumps(line)
    assert "\n" not in s
    assert len(s) < MAX_LINE_BYTES, f"bench line is {len(s)} bytes (limit {MAX_LINE_BYTES}): move detail to bench_detail.json"
    longest = max((len(x) for x in _strings(line)), default=0)
    assert longest <= MAX_STRING_CHARS, f"a string of {longest} characters in the bench line (limit {MAX_STRING_CHARS})"
    assert json.loads(s) == json.loads(json.dumps(json.loads(s)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    return s
