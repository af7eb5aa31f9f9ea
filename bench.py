#!/usr/bin/env python3
"""Benchmark of the per-step optimisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one ``optimizer.step(closure)`` of the reference loop
(optimization.py:175): VGG19 forward + Gram/content losses + backward to the
image + L-BFGS update.  Workload at N=1: BASELINE.json configs[1] (single
512x512 image, L-BFGS, VGG19 bf16 storage, --no-video); with --size 1024 it is
configs[2].  For N>1 every rank optimises its own independent content/style
pair (configs[3] pattern: no data-path collective; one RCCL all-gather of the
final images after the timed region) -> weak scaling.

Inputs are synthetic and resident in HBM before the timed region: U[0,1) RGB
images from the counter-hash PRNG, ImageNet-normalised; VGG19-topology weights
from the same PRNG (He-scaled) because the pretrained checkpoint cannot be
fetched here.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BF16_PEAK_TFLOPS = 2500.0     # dense MFMA peak, MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


class _Bar:
    def update(self, n=1):
        return None

    def set_postfix(self, *a, **k):
        return None

    def close(self):
        return None


def conv_flops(meta) -> float:
    """Algorithmic FLOPs of one conv op; a dgrad with the fused Gram-backward term (n = cin2 > 0 on a
    3x3 op: stv_conv_igemm_dual) also carries that term's 1x1 product."""
    op, H, W, cin, cout, taps, n = meta
    flops = 2.0 * max(taps, 1) * cin * cout * H * W
    if taps == 9 and n > 0:
        flops += 2.0 * n * cout * H * W
    return flops


_CFG_NAMES = {0: "8, 128, 4, 2", 1: "8, 64, 4, 2", 2: "4, 128, 1, 4", 3: "4, 64, 2, 2", 4: "4, 64, 2, 2", 5: "8, 64, 4, 2", 6: "4, 64, 2, 2", 7: "2, 64, 2, 2", 8: "1, 64, 1, 2",
              9: "16, 64, 4, 2", 10: "16, 64, 4, 2", 11: "2, 32, 2, 1", 12: "4, 32, 4, 1",
              13: "8, 64, 4, 2", 14: "16, 64, 4, 2", 15: "4, 64, 2, 2", 16: "2, 32, 2, 1", 17: "4, 32, 4, 1", 18: "16, 128, 4, 2"}
_CFG_KS = {0: "1, 3, false", 1: "1, 3, false", 2: "1, 3, false", 3: "1, 3, false", 4: "2, 3, false", 5: "1, 2, false", 6: "1, 2, false", 7: "2, 3, false",
           8: "2, 3, false", 9: "1, 3, false", 10: "1, 2, false", 11: "2, 3, false", 12: "2, 3, false",
           13: "1, 4, true", 14: "1, 4, true", 15: "1, 4, true", 16: "2, 4, true", 17: "2, 4, true", 18: "1, 2, false"}      # (tests/test_abi.py keeps these two tables as long as the library's list of tiles)


def kernel_group(meta, OP, dtype_code: int = 1, flags: int = 0) -> str | None:
    """Kernel (template instantiation) an op runs as - the names rocprofv3 --kernel-trace reports."""
    from style_transfer_visualizer_amd import _lib
    op, H, W, cin, cout, taps, _n = meta
    if op == OP["CONV"]:
        if taps == 9 and _lib.load().stv_conv_uses_ws(H, W, cin, cout, taps, dtype_code, 0, 1 if _n > 0 else 0, 0):
            return f"conv_ws_kernel<{cin}, {'true' if _n > 0 else 'false'}>"      # weight-stationary persistent kernel (Cin = 64 / 128; DG)
        # (a dgrad with the pooling backward in its epilogue has its own tune-table entry: stv.h STV_TUNE_ROUTE)
        cfg = _lib.load().stv_conv_config(H, W, cin, cout, 109 if (flags & _lib.POOL_ROUTE) else taps, dtype_code)
        elem = "unsigned short" if dtype_code == 1 else "float"
        if cfg < 0:
            return f"conv_direct_kernel<{elem}, {taps}>"
        relu = "true" if (flags & 1) else "false"                  # STV_RELU_IN: the ReLU-on-load instantiation
        if cfg not in _CFG_NAMES:                                  # a tile this table does not know yet: still a name
            return f"conv_igemm_kernel<tile {cfg}, {elem}, {taps}>, {relu}>"
        return f"conv_igemm_kernel<Cfg<{elem}, {_CFG_NAMES[cfg]}, {taps}, {_CFG_KS[cfg]}>, {relu}>"
    names = {OP["CONV_FIRST_FWD"]: "conv_first_fwd", OP["CONV_FIRST_DGRAD"]: "conv_first_dgrad",
             OP["POOL_FWD"]: "maxpool_fwd", OP["POOL_BWD"]: "maxpool_bwd", OP["GRAM_PARTIAL"]: "gram_partial",
             OP["GRAM_FINISH"]: "gram_finish", OP["GRAM_MULTI"]: "gram_multi (batched partial + finish)", OP["CONTENT_LOSS"]: "content_loss",
             OP["CONTENT_GRAD"]: "content_grad", OP["LOSS_COMBINE"]: "loss_combine",
             OP["RELU_FWD"]: "relu_fwd", OP["RELU_BWD"]: "relu_bwd"}
    return names.get(op)


def pmc_traffic(kernel: str, size: int) -> tuple[int | None, str | None]:
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary of this workload
    (profiles/r<NN>_pmc_hbm_<size>.json, newest round: FETCH_SIZE/WRITE_SIZE in separate passes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler itself."""
    import glob
    import re
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_{size}.json")))      # newest round last
    if not found:
        return None, None
    path = found[-1]

    def key(name: str) -> str:
        if "conv_ws_kernel" in name:                                         # conv_ws_kernel<CIN, DG, ...>: keyed by Cin and the backward flag
            mw = re.search(r"(conv_ws_kernel<\d+,\s*\w+)", name)
            return re.sub(r"\s+", "", mw.group(1)) if mw else re.sub(r"\s+", "", name)
        m = re.search(r"(\w+_kernel|\w+_c64)\W.*?((?:unsigned short|float)?[\d, ]*\d(?:,\s*(?:true|false))?)\s*>", name)
        if m is None:
            m2 = re.search(r"(conv_ws_kernel<\d+,\s*\w+)", name)       # conv_ws_kernel<CIN, DG, ...>: keyed by Cin and the backward flag
            return re.sub(r"\s+", "", m2.group(1)) if m2 else re.sub(r"\s+", "", name)
        inst = re.search(r">\s*,\s*(true|false)\s*>", name)                # conv_igemm_kernel<Cfg<...>, RELU>
        return re.sub(r"\s+", "", m.group(1) + "|" + m.group(2) + ("|" + inst.group(1) if inst else ""))
    try:
        data = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    for name, e in data.items():
        if key(name) == key(kernel):
            return int(e.get("hbm_bytes_per_launch", 0)) or None, os.path.relpath(path, ROOT)
    return None, None


def forward_bytes(sched, dtype_bytes: int) -> float:
    """SURVEY.md §8(d) byte model for forward+Gram: every tensor read once, written once."""
    total = 0.0
    for nd in sched.nodes:
        out_b = nd.dst.act.numel() * dtype_bytes
        if nd.kind == "conv_first":
            total += nd.dst.H * nd.dst.W * nd.cin * 4 + out_b + nd.wf.numel() * 4
        elif nd.kind == "conv":
            total += nd.src.act.numel() * dtype_bytes + out_b + nd.wf.numel() * 4   # weights counted fp32 as §8(d)
        else:
            total += nd.src.act.numel() * dtype_bytes + out_b
    for tap in sched.style_taps:
        total += tap.buf.act.numel() * dtype_bytes
    for tap in sched.content_taps:
        total += 2 * tap.buf.act.numel() * dtype_bytes
    return total


HISTORY_SIZE = 100


def run_gpu(args, rank: int, world: int, device: torch.device, size: int, steps: int, warmup: int, *,
            precision: str | None = None, profile: bool = True, prefill_history: bool = True):
    """One L-BFGS run of `warmup` untimed + `steps` timed optimisation steps through the real
    OptimizationRunner; returns elapsed seconds of the timed region (+ the kernel breakdown on rank 0)."""
    import torch.distributed as dist

    from style_transfer_visualizer_amd import _lib, config as stv_config
    from style_transfer_visualizer_amd import core_model, optimization, synthetic

    precision = precision or args.precision
    os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    # configs[1]/[2] are 300 / 500-step runs with history_size 100: from step 101 on every step sees a FULL
    # L-BFGS history (two sweeps over 2 x 100 vectors).  That steady state is what `value` rates, whatever
    # --warmup says: the history is filled by untimed steps first (set-up, like uploading the weights), then W
    # warm-up steps, then exactly K timed steps.
    prefill = max(0, HISTORY_SIZE - warmup) if prefill_history else 0
    warmup_total = prefill + warmup
    oc.steps = warmup_total + steps
    oc.init_method = "random"
    cfg.hardware.precision = precision
    cfg.video.create_video = False
    cfg.video.final_only = True
    cfg.output.log_every = 10
    torch.manual_seed(oc.seed + rank)
    torch.cuda.manual_seed_all(oc.seed + rank)
    content = synthetic.synthetic_image(2 * rank, size, size).to(device)
    style = synthetic.synthetic_image(2 * rank + 1, size, size).to(device)
    model, x, opt = core_model.prepare_model_and_input(content, style, device, oc, precision=precision)
    torch.cuda.synchronize(device)

    marks = {}

    def fence():
        torch.cuda.synchronize(device)
        marks["arrived"] = time.perf_counter()      # this rank's own clock, before it waits for the others
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        return time.perf_counter()

    def on_end(metrics):
        if metrics.step == warmup_total:
            marks["t0"] = fence()
        elif metrics.step == warmup_total + steps:
            marks["t1"] = fence()
            marks["t1_own"] = marks["arrived"]

    if warmup_total == 0:
        marks["t0"] = fence()
    runner = optimization.OptimizationRunner(
        model, x, cfg, optimizer=opt, progress_bar=_Bar(),
        callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    out, _history, _ = runner.run()
    elapsed = marks["t1"] - marks["t0"]
    elapsed_rank = marks["t1_own"] - marks["t0"]       # this rank's K steps alone (the line's time is the max over ranks, fences included)
    if world > 1:
        from style_transfer_visualizer_amd import parallel
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the only data collective of the job: gather the independent results (RCCL over xGMI)
        gathered = parallel.gather_results([(rank, out.detach().contiguous())], world)
        assert len(gathered) == world and all(g is not None for g in gathered)

    info = {"elapsed": elapsed, "elapsed_rank": elapsed_rank, "prefill": prefill}
    st = opt.device_state() if hasattr(opt, "device_state") else {}
    info["lbfgs"] = {k: st.get(k) for k in ("n_iter", "hist_len", "skip", "no_update")}
    if rank == 0 and hasattr(opt, "device_state"):
        # the optimizer update alone, at the history length the timed region ended with: 20 updates with the last
        # gradient inside one event pair.  Algorithmic bytes: two sweeps over the 2m history vectors + 8 vectors.
        from style_transfer_visualizer_amd import ops
        m = int(st.get("hist_len") or 0)
        n_el = x.numel()
        # on COPIES of the image, the gradient and the optimizer's state / history (the run's own state is what the
        # per-op profile below and the caller still use); two gradients alternate so that y = g - g_prev is not zero
        # (a repeated gradient takes the no-push path); the side stream starts behind the copies
        xs, gs = x.detach().clone(), x.grad.detach().clone()
        gs2 = gs * 0.995 + 1e-3 * gs.abs().mean() * torch.randn_like(gs)
        st_copy, work_copy = opt._dev_state.clone(), opt._work.clone()
        stream = torch.cuda.Stream(device=device)
        stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(23):
                if k == 3:
                    e0.record()
                ops.lbfgs_step(xs, gs if k % 2 == 0 else gs2, st_copy, work_copy, HISTORY_SIZE, HISTORY_SIZE, 1.0, compact=opt._compact)
            e1.record()
            e1.synchronize()
        torch.cuda.current_stream(device).wait_stream(stream)
        del st_copy, work_copy, xs, gs2
        ms = e0.elapsed_time(e1) / 20
        nbytes = (4 * m + 8) * n_el * 4
        info["lbfgs"].update(bytes=nbytes, ms=round(ms, 4), hbm_frac=round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             note="history read twice per step, exact fp32 pairs: the floor of this term is its HBM time")
    if rank == 0 and profile:
        # per-op device time of the fused step (HIP events on the launch stream), median of 5 passes
        eng = next(iter(model._engines.values()))
        # the closure ALONE: evaluated outside optimizer.step, nothing rides at the end of its schedule (inside the
        # runner the L-BFGS update is the schedule's last op, stv_op_t LBFGS_STEP; it is rated separately above)
        model.loss_and_grad(x, oc.style_w, oc.content_w, live_scores=True)
        prog = next(p for k, p in eng._programs.items() if k[0] == "fused" and k[-1] is None)
        OP = {n[3:]: getattr(_lib, n) for n in dir(_lib) if n.startswith("OP_")}
        side = torch.cuda.Stream(device=device)

        def replay_ms(run, n=30):
            """Whole-program device time in context: n captured-graph replays inside ONE event pair."""
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n
        with torch.cuda.stream(side):
            closure_ms = replay_ms(lambda: prog.run(True))
            xin = x.detach()
            fwd_gram_ctx_ms = replay_ms(lambda: eng.forward_losses(xin))
            prog.profile()
            # Per-op device time IN CONTEXT: every op once, in program order, inside its own event pair - its
            # operands are where the step leaves them (what `rocprofv3 --kernel-trace` averages over the run's
            # steps), not cache-hot from a back-to-back repeat.  Median of 7 passes.
            passes = [prog.profile(reps=1) for _ in range(7)]
            hot = [prog.profile(reps=8) for _ in range(3)]       # the same ops 8x back to back (cache-hot), for comparison
        # per op the MEDIAN of the passes: one disturbed pass (a clock dip, a neighbour's tail) must not rate a kernel
        ms = [sorted(p[i] for p in passes)[len(passes) // 2] for i in range(prog.n_ops)]
        ms_hot = [sorted(p[i] for p in hot)[len(hot) // 2] for i in range(prog.n_ops)]
        # An event pair around a single launch costs a few microseconds of its own.  Calibrated against the replayed
        # graph: the ops' true durations add up to the closure time minus the ~0.6 us between consecutive graph
        # kernels (tools/gap_report.py), so the per-op excess of the event-timed sum over that is the event cost.
        event_cost = max(0.0, (sum(ms) - (closure_ms - 0.0006 * prog.n_ops)) / prog.n_ops)
        ms = [max(t - event_cost, 0.25 * t) for t in ms]
        groups: dict = {}
        for meta, fl, t, th in zip(prog.op_meta, prog.op_flags, ms, ms_hot, strict=True):
            g = kernel_group(meta, OP, 1 if precision == "bf16" else 0, fl)
            e = groups.setdefault(g, {"ms": 0.0, "ms_hot": 0.0, "flops": 0.0, "launches": 0})
            e["ms"] += t
            e["ms_hot"] += th
            e["launches"] += 1
            if meta[0] == OP["CONV"]:
                e["flops"] += conv_flops(meta)
        if args.per_op:
            names = {v: k for k, v in OP.items()}
            with open(args.per_op, "a") as f:
                f.write(f"# size {size} precision {precision}\n")
                for meta, t in zip(prog.op_meta, ms, strict=True):
                    fl = conv_flops(meta) if meta[0] == OP["CONV"] else 0.0
                    f.write(f"{names.get(meta[0], meta[0]):18s} H{meta[1]:5d} W{meta[2]:5d} cin{meta[3]:4d} cout{meta[4]:4d} "
                            f"taps{meta[5]} n{meta[6]:9d}  {t * 1e3:9.1f} us  {fl / (t * 1e-3) / 1e12 if t > 0 else 0:8.1f} TF/s\n")
        dom_name, dom = max(((k, v) for k, v in groups.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
        peak = BF16_PEAK_TFLOPS if precision == "bf16" else FP32_PEAK_TFLOPS
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(dom_name, size) if precision == "bf16" else (None, None)
        info["roofline"] = {
            "bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "traffic_meaning": "fabric bytes incl. Infinity-Cache hits (FETCH_SIZE x2 + WRITE_SIZE, per launch): L2 fills and write-backs, an upper bound on HBM bytes",
            "launches_per_step": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 5),
            "avg_launch_ms_cache_hot": round(dom["ms_hot"] / dom["launches"], 5),
            "flop_per_launch": dom["flops"] / dom["launches"],
            "timing": ("HIP events around each op of the step program, in program order (operands as the step leaves them), "
                       f"minus the event pair's own cost ({event_cost * 1e3:.1f} us per op, calibrated so that the ops add up to "
                       "the replayed closure)"),
        }
        # forward + Gram/content losses = every op up to the score combine (SURVEY.md §8(d) byte model)
        # (timed in context above: replays of the forward-only program; the per-op pass below runs
        # every kernel 8x on cache-hot operands and is only used to rank and rate kernels)
        fwd_gram_ms = fwd_gram_ctx_ms
        dtype_bytes = 2 if precision == "bf16" else 4
        b_fwd = forward_bytes(eng.sched, dtype_bytes)
        total_flops = sum(e["flops"] for e in groups.values())
        step_ms = closure_ms
        info["breakdown_ms"] = {k: round(v["ms"], 4) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}
        info["fwd_gram"] = {"ms": round(fwd_gram_ms, 4), "algorithmic_MB": round(b_fwd / 1e6, 1),
                            "hbm_frac": round(b_fwd / (fwd_gram_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        info["closure"] = {"ms": round(step_ms, 4), "conv_gflop": round(total_flops / 1e9, 1),
                           "mfma_frac": round(total_flops / (step_ms * 1e-3) / 1e12 / peak, 4)}
    return info


def run_in_flight(device: torch.device, size: int, k: int, steps: int) -> dict:
    """K independent images on ONE GPU, each through its own OptimizationRunner on its own host thread and stream
    (what `main.style_transfer_batch` does with a rank's pairs): aggregate steps/s at a full L-BFGS history.  An extra
    leg - the headline `value` is one image per GPU, as BASELINE.json's configs are."""
    import threading

    from style_transfer_visualizer_amd import config as stv_config
    from style_transfer_visualizer_amd import core_model, optimization, synthetic
    gate = threading.Barrier(k)
    marks = [{} for _ in range(k)]
    setup = threading.Lock()
    errors: list = []

    def work(i: int) -> None:
        try:
            torch.cuda.set_device(device)
            cfg = stv_config.StyleTransferConfig.model_validate({})
            oc = cfg.optimization
            oc.steps, oc.init_method = HISTORY_SIZE + 10 + steps, "random"
            cfg.hardware.precision = "bf16"
            cfg.video.create_video, cfg.video.final_only = False, True
            with setup:
                torch.manual_seed(oc.seed + i)
                content = synthetic.synthetic_image(2 * i, size, size).to(device)
                style = synthetic.synthetic_image(2 * i + 1, size, size).to(device)
                model, x, opt = core_model.prepare_model_and_input(content, style, device, oc, precision="bf16")

            def on_end(m):
                if m.step == HISTORY_SIZE + 10:
                    torch.cuda.synchronize(device)
                    gate.wait(timeout=300)
                    marks[i]["t0"] = time.perf_counter()
                elif m.step == HISTORY_SIZE + 10 + steps:
                    torch.cuda.synchronize(device)
                    marks[i]["t1"] = time.perf_counter()
            optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar(),
                                            callbacks=optimization.OptimizationCallbacks(on_step_end=on_end)).run()
        except BaseException as exc:      # noqa: BLE001 - reported by the caller
            errors.append(exc)
            gate.abort()
    threads = [threading.Thread(target=work, args=(i,), name=f"stv-image-{i}") for i in range(k)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    span = max(m["t1"] for m in marks) - min(m["t0"] for m in marks)
    return {"images": k, "size": size, "steps_each": steps, "aggregate_steps_per_s": round(k * steps / span, 2),
            "ms_per_step_of_each_image": round(1e3 * span / steps, 4)}


def run_spatial(args, rank: int, world: int, device: torch.device) -> dict:
    """BASELINE configs[4]: ONE 3840x2160 image, Adam, row strips across the N GPUs (strong scaling)."""
    import torch.distributed as dist

    from style_transfer_visualizer_amd import core_model, spatial, synthetic
    os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
    H, W = 2160, 3840
    S, C = [0, 5, 10, 19, 28], [21]
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
    content = synthetic.synthetic_image(0, H, W).to(device)
    style = synthetic.synthetic_image(1, 1024, 1024).to(device)
    model = core_model.StyleContentModel(S, C, precision=args.precision).to(device)
    targets = model._engine_for(style).capture_style(style)
    torch.manual_seed(0)
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0)).to(device)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        return time.perf_counter()
    if args.spatial_mode == "recompute":
        shard = spatial.SpatialShard(model._layers(), S, C, content, targets, dtype=dtype, style_w=1e5, content_w=1.0)
        step = lambda: shard.adam_step(x, lr=1e-3)                       # noqa: E731
        mode = "recomputed 160-row halos"
        rows = [shard.c0, shard.c1, shard.e0, shard.e1]
    else:
        shard = spatial.HaloShard(model._layers(), S, C, content, targets, dtype=dtype, style_w=1e5, content_w=1.0)
        shard.set_image(x)
        step = lambda: shard.step("adam", lr=1e-3)                        # noqa: E731
        mode = f"1-row halo exchange before every 3x3 conv ({shard.exchanges_per_closure} per closure)"
        rows = [shard.c0, shard.c1]
    for _ in range(args.warmup):
        step()
    t0 = fence()
    for _ in range(args.steps):
        step()
    elapsed = fence() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"elapsed": elapsed, "rows": rows, "mode": mode, "scores": [float(v) for v in shard.last_scores.cpu()],
            "route": getattr(shard, "route", "eager")}


def cpu_baseline(size: int, threads: int) -> dict:
    """Reference algorithm on the host cores: the torch-CPU oracle (fp32), bounded sample."""
    from oracle import core_model_ref as ocm
    from oracle import optim_ref
    from style_transfer_visualizer_amd import synthetic

    torch.set_num_threads(threads)
    torch.manual_seed(0)
    weights = synthetic.synthetic_conv_weights(0)
    model = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), (0, 5, 10, 19, 28), (21,))
    content = synthetic.synthetic_image(0, size, size)
    style = synthetic.synthetic_image(1, size, size)
    model.set_targets(style, content)
    x = torch.randn(content.shape, generator=torch.Generator().manual_seed(0))
    opt = optim_ref.LbfgsRef(x.view(-1), lr=1.0)

    def closure():
        _s, _c, t, g = ocm.loss_and_grad(model, x, 1e5, 1.0)
        return t, g
    # ~10-15 s of CPU work per size on 16 cores (256^2: ~8 steps/s, 512^2: ~2, 1024^2: ~0.4)
    warm, timed = {256: (2, 60), 512: (2, 24)}.get(size, (1, 5))
    for _ in range(warm):
        opt.step(closure)
    t0 = time.perf_counter()
    for _ in range(timed):
        opt.step(closure)
    dt = time.perf_counter() - t0
    return {"value": round(timed / dt, 4), "unit": "steps/s", "cores": threads, "kind": "port", "cpu": host_cpu()["model"],
            "sample": f"{size}x{size} VGG19 fp32, torch-CPU oracle, {warm} warm-up + {timed} timed L-BFGS steps"}


def cpu_probe(size: int, threads: int, steps: int = 2) -> float:
    """Closures per second of the oracle at `threads` threads (1 warm-up + `steps` timed): thread-count choice only."""
    from oracle import core_model_ref as ocm
    from style_transfer_visualizer_amd import synthetic
    torch.set_num_threads(threads)
    weights = synthetic.synthetic_conv_weights(0)
    model = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), (0, 5, 10, 19, 28), (21,))
    content = synthetic.synthetic_image(0, size, size)
    model.set_targets(synthetic.synthetic_image(1, size, size), content)
    x = torch.randn(content.shape, generator=torch.Generator().manual_seed(0))
    ocm.loss_and_grad(model, x, 1e5, 1.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        ocm.loss_and_grad(model, x, 1e5, 1.0)
    return steps / (time.perf_counter() - t0)


def host_cpu() -> dict:
    """CPU model and the cores this process may use (physical cores, model stated)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = logical
    physical = None
    try:
        import psutil  # noqa: PLC0415
        physical = psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        physical = None
    # one thread per physical core, never more than this process is allowed to run on
    threads = max(1, min(allowed, physical or allowed))
    return {"model": model, "logical": logical, "physical": physical, "allowed": allowed, "threads": threads}


def _rccl_version() -> str:
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(p) for p in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001 - a version string must never take the bench line down
        return "unknown"


def _tile_info() -> dict:
    from style_transfer_visualizer_amd import _lib
    info = dict(_lib.tile_table_info)
    info["STV_CONV_TUNE"] = os.environ.get("STV_CONV_TUNE", "unset (table, no measuring)")
    return info


STEADY_FILL, STEADY_STEPS = 100, 60        # history_size = 100: after 100 steps every step runs at m = 100


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the additional legs (1024x1024, steady state at m=100, fp32 parity mode)")
    ap.add_argument("--per-op", default=None, help="append a per-op device-time table to this file")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a single GPU")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--spatial-mode", choices=["exchange", "recompute"], default="exchange",
                    help="--spatial partition: per-layer 1-row halo exchange (default) or recomputed 160-row halos")
    ap.add_argument("--spatial", action="store_true",
                    help="BASELINE configs[4] instead: one 3840x2160 image, Adam, row strips over the N GPUs")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves, as CHILD processes, before this
    # process has touched the GPU (never exec from a process that has initialised HIP).
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.run(cmd, check=False).returncode)

    from style_transfer_visualizer_amd import parallel
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} does not match WORLD_SIZE={world_env}: launch with "
                                   f"`python bench.py --gpus N` (spawns the ranks) or torchrun --nproc-per-node N"}))
        sys.exit(2)
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU: the HIP hot path has no CPU fallback"}))
        sys.exit(2)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(device)
    rank, _, world = parallel.init_distributed(args.dist_backend, device=device)

    if args.spatial:
        info = run_spatial(args, rank, world, device)
        if rank == 0:
            print(json.dumps({
                "metric": "optimization steps/sec, single 3840x2160 image", "value": round(args.steps / info["elapsed"], 3),
                "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * info["elapsed"] / args.steps, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
                "data": "synthetic",
                "config": {"workload": "single 3840x2160 image, Adam lr 1e-3, VGG19, row-strip partition "
                                       f"({info.get('mode', 'recomputed halos')}; BASELINE.json configs[4])",
                           "parallelism": f"spatial x{world}", "rank0_rows": info["rows"], "world_size": world,
                           "route": info["route"] + (" (closure replayed as one captured graph)" if info["route"] == "graph" else
                                                     " (closure issued segment by segment; STV_SPATIAL_GRAPH=1 opts a multi-rank run into the captured form)"),
                           "dist_backend": args.dist_backend},
                "scores": info["scores"]}))
        parallel.shutdown()
        return
    # N > 1 is BASELINE configs[3]: independent 1024x1024 pairs, one per GPU (no data-path collective).  The headline
    # of a multi-GPU line is therefore the 1024x1024 workload; 512x512 (configs[1]) runs as an extra leg on the same ranks.
    # N = 1 keeps configs[1] (512x512) as the headline and carries 1024x1024 as `extra_1024` / `roofline.step_1024`.
    head_size = args.size if (world == 1 or args.size != 512) else 1024
    info = run_gpu(args, rank, world, device, head_size, args.steps, args.warmup)
    per_rank = [info["elapsed_rank"]]
    if world > 1:
        import torch.distributed as dist
        t = torch.zeros(world, device=device, dtype=torch.float64)
        t[rank] = info["elapsed_rank"]
        dist.all_reduce(t)
        per_rank = [round(float(v), 6) for v in t.cpu()]
    extra = None
    extra_512 = None
    steady = None
    fp32 = None
    if world > 1 and head_size != args.size and not args.no_extra:
        e = run_gpu(args, rank, world, device, 512, args.steps, args.warmup, profile=False)
        extra_512 = {"workload": "single 512x512 image per GPU (BASELINE configs[1]), same ranks", "steps": args.steps,
                     "value": round(args.steps * world / e["elapsed"], 3), "ms_per_step": round(1e3 * e["elapsed"] / args.steps, 4)}
    if not args.no_extra and args.size == 512 and world == 1:
        k2, w2 = max(10, args.steps // 4), max(5, min(args.warmup, 100))
        e1024 = run_gpu(args, rank, world, device, 1024, k2, w2)
        extra = {"workload": "single 1024x1024 image (BASELINE configs[2])", "steps": k2, "warmup": w2,
                 "value": round(k2 / e1024["elapsed"], 3), "ms_per_step": round(1e3 * e1024["elapsed"] / k2, 4),
                 "roofline": e1024.get("roofline"), "fwd_gram": e1024.get("fwd_gram"), "closure": e1024.get("closure"),
                 "breakdown_ms": e1024.get("breakdown_ms"), "lbfgs": e1024.get("lbfgs")}
        # the reference's own arithmetic (fp32 parity mode): throughput at a full history AND its closure against the
        # fp32 matrix-core peak (v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate, 157.3 TFLOP/s)
        fp32 = run_gpu(args, rank, world, device, 512, STEADY_STEPS, STEADY_FILL, precision="fp32", prefill_history=False)
        steady = {"note": ("throughput with the L-BFGS history full (m = 100): the headline value and extra_1024 are "
                           "timed in that state themselves (history prefilled by untimed steps); the fp32 parity mode: "
                           f"{STEADY_FILL} untimed fill steps, then {STEADY_STEPS} timed steps through OptimizationRunner"),
                  "512x512_bf16": {"steps_per_s": round(args.steps / info["elapsed"], 2), "hist_len": info["lbfgs"].get("hist_len")},
                  "1024x1024_bf16": {"steps_per_s": extra["value"], "hist_len": extra["lbfgs"].get("hist_len")},
                  "512x512_fp32_parity_mode": {"steps_per_s": round(STEADY_STEPS / fp32["elapsed"], 2),
                                               "ms_per_step": round(1e3 * fp32["elapsed"] / STEADY_STEPS, 4),
                                               "hist_len": fp32["lbfgs"].get("hist_len"), "fill_steps": STEADY_FILL,
                                               "timed_steps": STEADY_STEPS, "precision": "fp32"}}

    in_flight = None
    if not args.no_extra and args.size == 512 and world == 1 and rank == 0:
        # several independent images on the one GPU (main.style_transfer_batch's default is 2 in flight)
        legs = [run_in_flight(device, 512, 2, 150), run_in_flight(device, 512, 3, 150), run_in_flight(device, 1024, 2, 60)]
        in_flight = {"note": ("independent images in flight on ONE GPU, one host thread + stream each: one image's L-BFGS update "
                              "(HBM-bound) overlaps another's closure (matrix-core / issue-bound); aggregate steps/s at a full "
                              "history; NOT the headline (BASELINE configs are one image per GPU)"),
                     "legs": legs}

    def step_summary(e: dict, steps_per_s: float) -> dict:
        """The three yardsticks of one workload, for the `roofline` object (the driver's record keeps that object)."""
        out = {"steps_per_s": round(steps_per_s, 2)}
        if e.get("fwd_gram"):
            out.update(fwd_gram_ms=e["fwd_gram"]["ms"], fwd_gram_hbm_frac=e["fwd_gram"]["hbm_frac"])
        if e.get("closure"):
            out.update(closure_ms=e["closure"]["ms"], closure_mfma_frac=e["closure"]["mfma_frac"])
        if e.get("lbfgs") and e["lbfgs"].get("ms") is not None:
            out.update(lbfgs_ms=e["lbfgs"]["ms"], lbfgs_hbm_frac=e["lbfgs"]["hbm_frac"])
        if e.get("roofline"):
            out.update(dominant_kernel=e["roofline"]["kernel"], dominant_frac=e["roofline"]["frac"],
                       dominant_traffic=e["roofline"]["traffic"])
        return out

    if rank == 0:
        total_steps = args.steps * world
        roof = info.get("roofline")
        if roof is not None:
            roof[f"step_{head_size}"] = step_summary(info, args.steps / info["elapsed"])
            if extra is not None:
                roof["step_1024"] = step_summary(e1024, extra["value"])
            if fp32 is not None and fp32.get("closure"):
                cl = fp32["closure"]
                roof["fp32_parity_mode"] = {
                    "steps_per_s": round(STEADY_STEPS / fp32["elapsed"], 2), "closure_ms": cl["ms"],
                    "tflops": round(cl["conv_gflop"] / cl["ms"], 2), "frac_of_157TF": cl["mfma_frac"],
                    "fwd_gram_ms": fp32["fwd_gram"]["ms"], "fwd_gram_hbm_frac": fp32["fwd_gram"]["hbm_frac"],
                    "note": "512x512, the reference's arithmetic (fp32 storage and sums): the mode that meets 1e-4 per pixel"}
            roof["peaks"] = {"hbm_GBs": HBM_PEAK_GBS, "bf16_mfma_TFLOPs": BF16_PEAK_TFLOPS, "fp32_mfma_TFLOPs": FP32_PEAK_TFLOPS}
        cfg_no = 1 if head_size == 512 else (3 if world > 1 else 2)
        line = {
            "metric": "optimization steps/sec at 512² and 1024², 1/2/4/8 MI355X",
            "value": round(total_steps / info["elapsed"], 3),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * info["elapsed"] / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"single {head_size}x{head_size} image per GPU, L-BFGS (max_iter=1, history 100), "
                             f"VGG19 {args.precision} storage / fp32 accumulate, fp32 Gram, --no-video "
                             f"(BASELINE.json configs[{cfg_no}]); weights: synthetic He-scaled"),
                "size": head_size, "images": world, "init_method": "random",
                "style_w": 1e5, "content_w": 1.0, "parallelism": f"replicas x{world} (independent images)",
                "lbfgs_history_prefill_steps": info.get("prefill", 0),
                "timed_region": "exactly --steps optimizer steps at a full L-BFGS history (m = 100), after --warmup untimed ones",
                "world_size": world,
                "per_rank_elapsed_s": per_rank,
                "collectives": ((f"RCCL {_rccl_version()} (torch.distributed backend 'nccl')" if args.dist_backend == "nccl"
                                 else f"{args.dist_backend} (rehearsal: ranks share one GPU, RCCL not used)")
                                + ": one all-gather of the final images after the timed region") if world > 1 else None,
                "single_gpu_reference": ("this line's workload at N = 1 is `extra_1024.value` / `roofline.step_1024.steps_per_s` of the "
                                         "`--gpus 1` line (whose headline is configs[1], 512x512)") if world > 1 and head_size == 1024 else None,
            },
            "tiles": _tile_info(),
            "roofline": roof,
            "fwd_gram": info.get("fwd_gram"),
            "closure": info.get("closure"),
            "breakdown_ms": info.get("breakdown_ms"),
            "lbfgs": info.get("lbfgs"),
        }
        if extra_512 is not None:
            line["extra_512"] = extra_512
        if steady is not None:
            line["steady_state"] = steady
        if extra is not None:
            line["extra_1024"] = extra
        if in_flight is not None:
            line["images_in_flight"] = in_flight
            if roof is not None:            # the driver's record keeps `roofline`: the aggregate figures go there too
                roof["images_in_flight_steps_per_s"] = {f"{g['size']}x{g['size']} x{g['images']}": g["aggregate_steps_per_s"] for g in in_flight["legs"]}
        if world == 1 and not args.no_cpu_baseline:
            cpu = host_cpu()
            # BASELINE.md §4 asks for the node's physical cores; on a many-core host the reference's CPU path is
            # not fastest there (small convolutions, 128 threads), so a short probe also tries 64, 32, 16 and 8 threads and
            # the timed sample runs on whichever is fastest - every probed count is reported.
            probe = {}
            for nt in sorted({cpu["threads"], *(min(k, cpu["threads"]) for k in (64, 32, 16, 8))}, reverse=True):
                probe[nt] = cpu_probe(args.size, nt)
            threads = max(probe, key=probe.get)
            cpu["probe_steps_per_s"] = {str(k): round(v, 3) for k, v in probe.items()}
            line["host_cpu"] = cpu
            line["cpu_baseline"] = cpu_baseline(args.size, threads)
            if not args.no_extra and args.size == 512:      # the other two sizes beside it
                line["cpu_baseline_other_sizes"] = {"256x256": cpu_baseline(256, threads),
                                                    "1024x1024": cpu_baseline(1024, threads)}
        print(json.dumps(line))
    parallel.shutdown()


if __name__ == "__main__":
    main()
