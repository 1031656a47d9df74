#!/usr/bin/env python3
"""Headline benchmark: focal-stacks/sec of the depth-from-focus forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16x3|fp16|bf16]

A "step" is one pass of the hot path (DFF_net.forward, reference DEN.py:74-127, through
libdffw.so) over one batch of synthetic 10-slice 3x256x256 focal stacks per GPU — BASELINE.json's
config "DefocusNet-shape 10-slice 256x256 stacks, batch=32, 1xMI355X" (config 3; config 4 is the
same per-GPU batch on 8 GPUs).  Inputs are resident in HBM before the timed region.  For N > 1 one
rank runs per GPU -- launched by torch.distributed.run, or, when bench.py is invoked plainly with
--gpus N, by bench.py itself (N fresh child processes) -- every rank processes its own 32 stacks
(weak scaling, no data-path collective) and the per-rank depth maps are collected with one RCCL
all-gather inside the timed step (its bus GB/s is reported under "allgather").

`--workload e2e` runs BASELINE.json's config 5 instead (not the default line): the End_to_End variant —
alignment network + FOV warp + DFF_net (End_to_End/End_to_End.py) — on 10-slice 480x640 stacks, batch 8.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (by summed time) of one profiled forward: algorithmic FLOPs per
                launch / HIP-event duration per launch against the dense MFMA peak of the
                instruction type issued (2.5 PFLOP/s bf16/f16; split-bf16 issues 3 MFMAs per
                algorithmic product, so its ceiling is 1/3 of that)
  cpu_baseline  the oracle (oracle/cpu_ref.py = the reference's PyTorch-CPU arithmetic, restated)
                timed on this box's host cores on a bounded sample
  parity        rel-L2 / RMSE of this run's pred3 for stack 0 against the oracle
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dffinthewild_amd import dist as ddist  # noqa: E402
from dffinthewild_amd import graph, synth  # noqa: E402

GFLOP_PER_STACK = 61.317          # 10x256x256, 2*MAC over the 70 convs
PEAK_MFMA_TFLOPS = 2500.0         # dense bf16/f16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def build_model(precision, device, workload="depth"):
    if workload == "e2e":
        from dffinthewild_amd.End_to_End import Network
        entries = list(graph.param_entries(graph.e2e_convs()))
    else:
        from dffinthewild_amd.Depth_Estimation_Network import Network
        entries = list(graph.param_entries(graph.dff_net_convs()))
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(entries, seed=0, profile="smooth").items()}
    model = Network(precision=precision)
    model.load_state_dict(sd)
    return model.to(device).eval(), sd


def roofline_from_profile(model, inputs, device, precision="bf16x3"):
    eng = model._engine_on(device)
    eng.profile(True)
    with torch.no_grad():
        model(*inputs)
    torch.cuda.synchronize(device)
    rows = eng.profile_collect()
    eng.profile(False)
    agg = {}
    for kernel, layer, flops, nbytes, ms in rows:
        a = agg.setdefault(kernel, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
        a["launches"] += 1
        a["flops"] += flops
        a["bytes"] += nbytes
        a["ms"] += ms
    total_ms = sum(a["ms"] for a in agg.values())
    conv = {k: a for k, a in agg.items() if "::conv_" in k}   # the MFMA implicit-GEMM kernels (conv_tile / conv_igemm)
    dom_name, dom = max(conv.items(), key=lambda kv: kv[1]["ms"])
    conv_flops = sum(a["flops"] for a in conv.values())
    conv_ms = sum(a["ms"] for a in conv.values())

    def fractions(a):
        tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
        gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
        return tf, gbs, tf / PEAK_MFMA_TFLOPS, gbs / PEAK_HBM_GBS

    tf, gbs, f_mfma, f_hbm = fractions(dom)
    # which roof binds the kernel: the larger of its two minimum times, MFMA time counted with the instruction issues the
    # arithmetic mode really pays per algorithmic product (split-bf16: 3) -- SURVEY.md 8d
    issue = 3.0 if precision == "bf16x3" else 1.0
    t_mfma = dom["flops"] * issue / (PEAK_MFMA_TFLOPS * 1e12)
    t_hbm = dom["bytes"] / (PEAK_HBM_GBS * 1e9)
    hbm_bound = t_hbm > t_mfma
    roof = {
        "bound": "hbm" if hbm_bound else "mfma", "kernel": dom_name,
        "achieved": round(gbs if hbm_bound else tf, 2), "peak": PEAK_HBM_GBS if hbm_bound else PEAK_MFMA_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(f_hbm if hbm_bound else f_mfma, 4), "traffic": None,
        "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
        "gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
        "algorithmic_gb_per_launch": round(dom["bytes"] / dom["launches"] / 1e9, 4),
        "tflops": round(tf, 2), "frac_mfma": round(f_mfma, 4), "algorithmic_gbs": round(gbs, 1), "frac_hbm": round(f_hbm, 4),
        "mfma_issues_per_product": issue, "frac_of_issue_ceiling": round(f_mfma * issue, 4),
        "arithmetic_intensity_flop_per_byte": round(dom["flops"] / dom["bytes"], 1),
        "bound_rule": "mfma if algorithmic FLOPs x MFMA issues per product / 2.5 PFLOP/s >= algorithmic bytes / 8 TB/s, else hbm; "
                      "frac = algorithmic work / measured time / peak (so a split-bf16 kernel tops out at 1/3 of the dense MFMA peak)",
        "share_of_forward_time": round(dom["ms"] / total_ms, 3),
        "all_conv_kernels": {"achieved": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                             "frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4),
                             "share_of_forward_time": round(conv_ms / total_ms, 3)},
        # the dominant kernel's launches one by one (since round 6 one instantiation serves layers of 8x different size: `frac` above is their total
        # work over their total time)
        "launches_of_kernel": [{"layer": lname, "ms": round(ms, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 1),
                                "frac_mfma": round(fl / (ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4)}
                               for k, lname, fl, _, ms in rows if k == dom_name and ms > 0],
        "profiled_forward_ms": round(total_ms, 3), "n_launches": len(rows),
        "durations": "HIP events around every launch of one extra forward after the timed region, on the launch stream; in this "
                     "profiling mode the engine runs the pyramid's three scales one after the other (side by side, as in the timed "
                     "steps, small kernels share the chip and a launch's duration is not the kernel's own)",
    }
    # what this box sustains (the chip clocks to its power budget: a register-resident MFMA loop and a streaming copy, timed with
    # HIP events like the kernels above): the same fractions against the measured ceilings
    try:
        if os.environ.get("DFFW_NO_PROBE") == "1":       # (profiler runs: keep the probe kernels out of the kernel statistics)
            raise RuntimeError("skipped (DFFW_NO_PROBE=1)")
        from dffinthewild_amd import engine as _eng
        m_tf, h_gbs = _eng.probe_peaks(device.index or 0)
        roof["measured_ceilings"] = {"mfma_tflops": round(m_tf, 1), "hbm_copy_gbs": round(h_gbs, 1),
                                     "frac_of_measured": round((gbs / h_gbs) if hbm_bound else (tf * issue / m_tf), 4),
                                     "note": "v_mfma_f32_16x16x32_bf16 back to back out of registers on every SIMD / float4 streaming (copy or read, the better), on this GPU, "
                                             "just now; frac_of_measured = this kernel's rate (x MFMA issues per product) over that"}
    except Exception as exc:   # noqa: BLE001 -- a failed probe must not cost the bench line
        roof["measured_ceilings"] = {"error": str(exc)}
    # whole forward against its layer-by-layer roofline: sum over launches of max(MFMA time, HBM time) with each launch's
    # algorithmic FLOPs (x3 MFMA issue in the split-bf16 mode) and algorithmic bytes ("the exact ceiling is the
    # per-layer sum of max(.,.)")
    bound_ms = sum(max(flops * issue / (PEAK_MFMA_TFLOPS * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)) for _, _, flops, nbytes, _ in rows) * 1e3
    roof["forward_vs_layerwise_roofline"] = {"bound_ms": round(bound_ms, 3), "measured_ms": round(total_ms, 3), "frac": round(bound_ms / total_ms, 4),
                                             "definition": "sum over the launches of max(algorithmic FLOPs x MFMA issues per product / dense peak, "
                                                           "algorithmic bytes / 8 TB/s) / sum of the launches' measured durations"}
    # the north star's graded subset: the 3-D cost-aggregation convs (SPP_module, confidence,
    # dres0, deconv_1..3, dres2..4; 48.685 GF per 10x256x256 stack)
    agg3d = [(fl, ms) for _, layer, fl, _, ms in rows
             if any(layer.startswith("DFF_net." + p) for p in ("SPP_module", "confidence", "dres0", "deconv_", "dres2", "dres3", "dres4"))]
    if agg3d:
        gfl, gms = sum(f for f, _ in agg3d), sum(m for _, m in agg3d)
        gtf = gfl / (gms * 1e-3) / 1e12
        roof["aggregation_3d_convs"] = {"gflop": round(gfl / 1e9, 2), "ms": round(gms, 3), "achieved": round(gtf, 2), "unit": "TFLOP/s",
                                        "frac": round(gtf / PEAK_MFMA_TFLOPS, 4), "frac_of_issue_ceiling": round(gtf * issue / PEAK_MFMA_TFLOPS, 4),
                                        "launches": len(agg3d), "share_of_forward_time": round(gms / total_ms, 3)}
    # the heaviest kernel that is MFMA-bound by the same criterion (the 3x3x3 aggregation convs of the north star)
    mf = {k: a for k, a in conv.items() if a["flops"] * issue / (PEAK_MFMA_TFLOPS * 1e12) >= a["bytes"] / (PEAK_HBM_GBS * 1e9)}
    if mf:
        mk, ma = max(mf.items(), key=lambda kv: kv[1]["ms"])
        mtf, mgbs, mfm, mfh = fractions(ma)
        roof["top_mfma_bound_kernel"] = {"kernel": mk, "achieved": round(mtf, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(mfm, 4),
                                         "frac_of_split_bf16_ceiling": round(mfm * 3, 4) if precision == "bf16x3" else None,
                                         "launches": ma["launches"], "avg_launch_ms": round(ma["ms"] / ma["launches"], 4),
                                         "share_of_forward_time": round(ma["ms"] / total_ms, 3)}
    # the HBM-bound side of the forward: every launch whose algorithmic bytes / 8 TB/s exceed its MFMA issue time, and the
    # heaviest kernel among them (ALL kernels, not only the convs: the fused SRD blocks, the pools and the regression heads are HBM-bound)
    hb = {k: a for k, a in agg.items() if a["bytes"] / (PEAK_HBM_GBS * 1e9) > a["flops"] * issue / (PEAK_MFMA_TFLOPS * 1e12)}
    hb_rows = [(fl, by, ms) for _, _, fl, by, ms in rows if by / (PEAK_HBM_GBS * 1e9) > fl * issue / (PEAK_MFMA_TFLOPS * 1e12)]
    if hb_rows:
        hms, hby = sum(m for _, _, m in hb_rows), sum(b for _, b, _ in hb_rows)
        roof["hbm_bound_share_of_forward_time"] = round(hms / total_ms, 3)
        roof["hbm_bound_launches"] = {"launches": len(hb_rows), "ms": round(hms, 3), "algorithmic_gb": round(hby / 1e9, 2),
                                      "achieved": round(hby / (hms * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(hby / (hms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                      "definition": "launch-by-launch: algorithmic bytes / 8 TB/s > algorithmic FLOPs x MFMA issues per product / 2.5 PFLOP/s"}
    top_hbm = None
    if hb:
        hk, ha = max(hb.items(), key=lambda kv: kv[1]["ms"])
        _, hgbs, _, hfh = fractions(ha)
        top_hbm = hk
        roof["top_hbm_bound_kernel"] = {"kernel": hk, "achieved": round(hgbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(hfh, 4),
                                        "launches": ha["launches"], "avg_launch_ms": round(ha["ms"] / ha["launches"], 4),
                                        "algorithmic_gb_per_launch": round(ha["bytes"] / ha["launches"] / 1e9, 4),
                                        "share_of_forward_time": round(ha["ms"] / total_ms, 3), "traffic": None, "traffic_over_algorithmic": None}
    # HBM traffic of the dominant kernel: measured offline with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
    # cannot be collected from inside this process) and committed under profiles/; reported only when it
    # was measured for this very kernel instantiation, else null
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")), reverse=True):
            with open(path) as f:
                tr = json.load(f)
            src = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes; gfx950 x2 fetch correction)"
            if top_hbm and roof["top_hbm_bound_kernel"]["traffic"] is None:
                for ent in [tr] + tr.get("other_kernels", []):
                    if ent.get("kernel") == top_hbm:
                        th = roof["top_hbm_bound_kernel"]
                        th["traffic"] = round(ent["hbm_bytes_per_launch"])
                        th["traffic_over_algorithmic"] = round(ent["hbm_bytes_per_launch"] / (th["algorithmic_gb_per_launch"] * 1e9), 3)
                        th["traffic_source"] = src
            if tr.get("kernel") == dom_name and roof["traffic"] is None:
                roof["traffic"] = round(tr["hbm_bytes_per_launch"])
                roof["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / (dom["bytes"] / dom["launches"]), 3)
                roof["traffic_source"] = src
            if roof["traffic"] is not None and (not top_hbm or roof["top_hbm_bound_kernel"]["traffic"] is not None):
                break
    except (OSError, ValueError, KeyError):
        pass
    per_kernel = {k: {"launches": a["launches"], "ms": round(a["ms"], 3),
                      "tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2) if a["flops"] else None,
                      "gbs": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)} for k, a in sorted(agg.items())}
    return roof, per_kernel, rows


def relative_fovs(B, N):
    """Relative field of view per slice, decreasing to 1 at the last (reference) slice (Test_dataloader.py:56-70)."""
    f = 1.0 + 0.06 * torch.arange(N - 1, -1, -1, dtype=torch.float32) / max(N - 1, 1)
    return f.reshape(1, 1, N, 1, 1).repeat(B, 1, 1, 1, 1)


def cpu_baseline_e2e(sd, seconds, H, W, batch=2):
    """Oracle End_to_End forward (oracle/cpu_ref.py e2e_forward) on `batch` stacks of the same shape, same
    thread policy as cpu_baseline."""
    from oracle import cpu_ref
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 32)
    torch.set_num_threads(cores)
    FS = torch.from_numpy(synth.focal_stack(batch, 10, H, W, seed=1000))
    fd = torch.from_numpy(synth.focus_dists(batch, 10, 1, 1))
    fov = relative_fovs(batch, 10)
    with torch.no_grad():
        ref = cpu_ref.e2e_forward(sd, FS, fd, fov)
        times = []
        t_end = time.time() + seconds
        while len(times) < 1 or (time.time() < t_end and len(times) < 10):
            t0 = time.perf_counter()
            cpu_ref.e2e_forward(sd, FS, fd, fov)
            times.append(time.perf_counter() - t0)
    best = min(times)
    base = {"value": round(batch / best, 3), "unit": "stacks/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} End_to_End forwards of one batch of {batch} 10x3x{H}x{W} stacks after 1 warm-up, best of "
                      f"(mean {sum(times)/len(times):.2f} s per batch); oracle/cpu_ref.py e2e_forward; {avail} logical cores "
                      f"available, {cores} threads used"}
    return base, ref


def physical_cores(default):
    """Physical cores this process may run on (unique (package, core) pairs of /proc/cpuinfo within the affinity mask)."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, pkg = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    cpu = int(line.split(":")[1])
                elif line.startswith("physical id"):
                    pkg = int(line.split(":")[1])
                elif line.startswith("core id") and cpu in allowed:
                    seen.add((pkg, int(line.split(":")[1])))
        return len(seen) or default
    except (OSError, ValueError, AttributeError):
        return default


def cpu_baseline(sd, seconds, batch=8):
    """Oracle forward on a bounded sample (one batch of `batch` 10x256x256 stacks, repeated) on this box's
    host cores.  PyTorch-CPU stops scaling (and degrades) past ~32 threads for these small convs
    (tools/cpu_threads_sweep.py: 32 threads is the fastest setting on the 256-core EPYC host), so the
    thread count is min(cores available, 32); `cores` in the JSON is what was actually used."""
    from oracle import cpu_ref
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = physical_cores(avail)
    FS = torch.from_numpy(synth.focal_stack(batch, 10, 256, 256, seed=1000))
    fd = torch.from_numpy(synth.focus_dists(batch, 10, 256, 256))
    # two thread policies share the time budget: all physical cores and 32 threads (the optimum of the
    # sweep in profiles/*cpu_threads_sweep*: PyTorch-CPU degrades past ~32 threads on these small convs); the faster one is
    # the reported baseline, both are named in `sample`
    policies = sorted({min(avail, 32), min(avail, phys)})
    tried = {}
    ref = None
    with torch.no_grad():
        for cores_k in policies:
            torch.set_num_threads(cores_k)
            r = cpu_ref.dff_forward(sd, FS, fd)          # warm-up, also the parity reference for the first stacks
            ref = ref if ref is not None else r
            ts = []
            t_end = time.time() + seconds / len(policies)
            while len(ts) < 1 or (time.time() < t_end and len(ts) < 10):
                t0 = time.perf_counter()
                cpu_ref.dff_forward(sd, FS, fd)
                ts.append(time.perf_counter() - t0)
            tried[cores_k] = ts
    cores = min(tried, key=lambda k: min(tried[k]))
    times = tried[cores]
    best = min(times)
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    base = {"value": round(batch / best, 3), "unit": "stacks/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} forwards of one batch of {batch} 10x3x256x256 stacks after 1 warm-up, best of "
                      f"(mean {sum(times)/len(times):.2f} s per batch); oracle/cpu_ref.py = the reference's PyTorch-CPU "
                      f"fp32 arithmetic restated; {avail} logical / {phys} physical cores available, {cores} threads used "
                      f"(best of the policies tried: " + ", ".join(f"{k} threads {batch / min(v):.2f} stacks/s" for k, v in sorted(tried.items())) + ")",
            "cpu": model_name}
    return base, ref


def time_config(model, inputs, steps, warmup):
    """`steps` forwards of `model(*inputs)` after `warmup`, bracketed by device synchronisations; seconds per step."""
    with torch.no_grad():
        for _ in range(warmup):
            model(*inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(*inputs)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def other_configs(depth_model, precision, device):
    """The BASELINE.json configs the headline line does not cover, as short in-process runs after the timed region (same engine,
    same synthetic weights; inputs resident in HBM): config 1 (one 10x3x256x256 stack), config 2 (one 5x3x224x224 stack) on the
    depth network, config 5 (End_to_End, 8 stacks of 10x3x480x640)."""
    out = {}
    for key, (B, N, H, W, steps, warmup) in {"config1_b1_10x256": (1, 10, 256, 256, 100, 20), "config2_b1_5x224": (1, 5, 224, 224, 100, 20)}.items():
        FS = torch.from_numpy(synth.focal_stack(B, N, H, W, seed=1000)).to(device)
        fd = torch.from_numpy(synth.focus_dists(B, N, H, W)).to(device)
        sec = time_config(depth_model, (FS, fd), steps, warmup)
        out[key] = {"stacks_per_s": round(B / sec, 1), "ms_per_step": round(sec * 1e3, 4), "steps": steps, "warmup": warmup,
                    "workload": f"DFF_net forward, {B} stack of {N}x3x{H}x{W}, dense focus_dists"}
    B, N, H, W, steps, warmup = 8, 10, 480, 640, 10, 3
    e2e_model, _ = build_model(precision, device, "e2e")
    FS = torch.from_numpy(synth.focal_stack(B, N, H, W, seed=1000)).to(device)
    fd = torch.from_numpy(synth.focus_dists(B, N, 1, 1)).to(device)
    sec = time_config(e2e_model, (FS, fd, relative_fovs(B, N).to(device)), steps, warmup)
    out["config5_e2e_b8_480x640"] = {"stacks_per_s": round(B / sec, 1), "ms_per_step": round(sec * 1e3, 3), "steps": steps, "warmup": warmup,
                                     "workload": f"End_to_End forward (alignment network + FOV warp + DFF_net), {B} stacks of {N}x3x{H}x{W}, broadcast focus_dists"}
    del e2e_model
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without an outer launcher: N child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run would set them), rendezvous on 127.0.0.1.  Rank 0's stdout (the JSON line) is passed
    through; the exit code is the worst child's."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    # rank 0's stdout goes to a file, not a pipe: nobody has to drain it while the children are polled
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # poll: the first rank that fails (or an overall timeout) takes its siblings down instead of leaving them -- and this
        # parent -- parked in the rendezvous or in a collective for good
        deadline = time.time() + float(os.environ.get("DFFW_BENCH_TIMEOUT_S", "1500"))
        codes = [None] * n
        failed = None
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
                    if codes[i] not in (None, 0) and failed is None:
                        failed = i
            if failed is not None or time.time() > deadline:
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            codes[i] = p.wait(timeout=10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[i] = p.wait()
                if failed is None:
                    sys.stderr.write("bench.py: ranks did not finish in time, terminated\n")
                    codes = [c if c else 124 for c in codes]
                else:
                    sys.stderr.write(f"bench.py: rank {failed} exited with {codes[failed]}, the other ranks were terminated\n")
                break
            time.sleep(0.05)
        out0.seek(0)
        out = out0.read()
    # rank 0's stdout also carries the communication library's banner lines: pass on the JSON line only (the rest to stderr)
    for line in out.decode(errors="replace").splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if failed is not None:
        return abs(codes[failed]) or 1
    return max(abs(c) for c in codes)


def measure_allgather(local, world, iters=20):
    """The step's one collective on its own: RCCL all-gather of this rank's pred3 maps, timed over `iters` calls between
    device synchronisations (max over ranks).  bus GB/s = bytes every rank receives from the others / time."""
    total = world * local.shape[0]          # equal shards: exactly one all_gather_into_tensor per call, as in the timed step
    for _ in range(3):
        ddist.all_gather_depth(local, total=total)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        ddist.all_gather_depth(local, total=total)
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=local.device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    sec = float(t.item())
    nbytes = local.numel() * local.element_size()
    return {"collective": f"all_gather_into_tensor (backend {torch.distributed.get_backend()}" + (" = RCCL over xGMI)" if torch.distributed.get_backend() == "nccl" else ", test rig)"),
            "ranks": torch.distributed.get_world_size(),
            "bytes_per_rank": nbytes, "result_bytes": nbytes * world, "us_per_call": round(sec * 1e6, 1),
            "algbw_GBs": round(nbytes * world / sec / 1e9, 2), "busbw_GBs": round(nbytes * (world - 1) / sec / 1e9, 2),
            "sample": f"{iters} back-to-back calls after 3 warm-ups, outside the timed steps (each timed step also contains one)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("depth", "e2e"), default="depth",
                    help="depth: DFF_net forward (BASELINE config 3/4, the headline); e2e: End_to_End forward (config 5)")
    ap.add_argument("--batch", type=int, default=None, help="stacks per GPU per step (default 32; 8 for --workload e2e)")
    ap.add_argument("--slices", type=int, default=10)
    ap.add_argument("--size", type=int, default=None, help="square stack size (default 256)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--precision", default=os.environ.get("DFFW_PRECISION", "bf16x3"))
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--input", choices=("fp32", "u8"), default="fp32",
                    help="fp32: the reference's tensor contract (B,3,N,H,W) float32 (default, the headline); u8: the raw uint8 "
                         "(B,N,H,W,3) stack a loader holds before /127.5-1, normalised inside the stem kernel (Network.forward_raw)")
    ap.add_argument("--dump-layers", default=None, help="write the per-launch profile table to this file")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 1, 2 and 5 (`configs` key)")
    ap.add_argument("--sustain-seconds", type=float, default=4.0,
                    help="after the K timed steps, keep stepping until this many seconds have been timed in total and report it as `sustained`")
    args = ap.parse_args()

    rank, local_rank, world = ddist.env_world()
    if world == 1 and args.gpus > 1:
        # invoked plainly (no torch.distributed.run around it): start one fresh process per GPU ourselves.  Nothing in
        # this parent has touched the GPU yet, and the children are new processes (spawned, not exec'ed over this one).
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # DFFW_BENCH_ONE_DEVICE=1 (test rig for the multi-process path on a one-GPU box): every rank on cuda:0, gloo instead of RCCL
    one_dev = os.environ.get("DFFW_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        ddist.init_process_group("gloo" if one_dev else "nccl", set_device=not one_dev)

    e2e = args.workload == "e2e"
    B, N = args.batch or (8 if e2e else 32), args.slices
    Hh = args.height or args.size or (480 if e2e else 256)
    Ww = args.width or args.size or (640 if e2e else 256)
    S = Hh if Hh == Ww else None
    model, sd = build_model(args.precision, device, args.workload)
    FS = torch.from_numpy(synth.focal_stack(B, N, Hh, Ww, seed=1000 + rank)).to(device)
    if e2e:
        fd = torch.from_numpy(synth.focus_dists(B, N, 1, 1)).to(device)   # (B,10,1,1), as Test_dataloader.py:52-54
        inputs = (FS, fd, relative_fovs(B, N).to(device))
    else:
        fd = torch.from_numpy(synth.focus_dists(B, N, Hh, Ww)).to(device)  # dense tile, as test_Dataloader.py:24
        inputs = (FS, fd)

    raw_u8 = None
    if args.input == "u8":
        if e2e:
            raise SystemExit("--input u8 is implemented for --workload depth")
        # same synthetic stack quantised to the loaders' source format: uint8 (B,N,H,W,3)
        raw_u8 = ((FS + 1.0) * 127.5).round().clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 4, 1).contiguous()

    def step():
        with torch.no_grad():
            outs = model.forward_raw(raw_u8, fd, "NHWC") if raw_u8 is not None else model(*inputs)
            # every rank runs B stacks: the total is known, so the step's only collective is ONE all_gather_into_tensor (no size
            # negotiation, no host sync)
            gathered = ddist.all_gather_depth(outs[3], total=world * B) if world > 1 else outs[3]
        return outs, gathered

    for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs, gathered = step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    stacks = world * B * args.steps
    value = stacks / elapsed
    # the K steps of the contract are `value`; the same step repeated until >= --sustain-seconds have been timed (single GPU:
    # the default K = 10 is a 0.1 s region) is reported beside it as `sustained`
    sustained = None
    if world == 1 and args.sustain_seconds > 0 and elapsed < args.sustain_seconds:
        more = int((args.sustain_seconds - elapsed) / (elapsed / args.steps)) + 1
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(more):
            step()
        torch.cuda.synchronize(device)
        el2 = time.perf_counter() - t1
        sustained = {"steps_effective": args.steps + more, "seconds": round(elapsed + el2, 3),
                     "value": round(B * (args.steps + more) / (elapsed + el2), 2), "unit": "stacks/s",
                     "ms_per_step": round((elapsed + el2) / (args.steps + more) * 1e3, 3)}
    allgather = measure_allgather(outs[3], world) if world > 1 else None

    result = None
    if rank == 0:
        scale = (N * Hh * Ww) / (10 * 256 * 256)
        result = {
            "metric": "focal-stacks/sec (10x3x480x640, End_to_End)" if e2e else "focal-stacks/sec (10x3x256x256)",
            "value": round(value, 2), "unit": "stacks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (split-bf16 MFMA operands hi+lo, fp32 accumulate, fp32-accurate)",
                      "fp16": "f16 (MFMA, fp32 accumulate)", "bf16": "bf16 (MFMA, fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": (f"End_to_End forward (alignment network + FOV warp + DFF_net), {N}-slice 3x{Hh}x{Ww} focal "
                                    f"stacks, batch {B} per GPU (BASELINE.json config 5), broadcast focus_dists, synthetic "
                                    f"weights seed 0") if e2e else
                                   (f"DFF_net forward, {N}-slice 3x{Hh}x{Ww} focal stacks, batch {B} per GPU "
                                    f"(BASELINE.json config 3{'/4' if world > 1 else ''}), dense focus_dists, "
                                    f"synthetic weights seed 0"),
                       "batch_per_gpu": B, "global_batch": B * world, "slices": N, "height": Hh, "width": Ww,
                       "parallelism": (f"batch-sharded x{world}, RCCL all-gather of pred3" if not one_dev else f"TEST RIG: {world} ranks on one GPU, gloo") if world > 1 else "single GPU",
                       "precision": args.precision, "input": "uint8 (B,N,H,W,3), normalised in the stem kernel" if raw_u8 is not None
                       else "float32 (B,3,N,H,W), the reference's tensor contract"},
            "forward_tflops_algorithmic": round(value * GFLOP_PER_STACK * scale / 1e3, 2),   # DFF_net's convs only
        }
        if allgather:
            result["allgather"] = allgather
        if sustained:
            result["sustained"] = sustained
    if rank == 0 and not args.no_roofline and raw_u8 is None:
        roof, per_kernel, rows = roofline_from_profile(model, inputs, device, args.precision)
        result["roofline"] = roof
        result["kernels"] = per_kernel
        if args.dump_layers:
            with open(args.dump_layers, "w") as f:
                f.write("kernel\tlayer\tgflop\talg_MB\tms\ttflops\talg_GBs\n")
                for k, l, fl, by, ms in rows:
                    f.write(f"{k}\t{l}\t{fl/1e9:.3f}\t{by/1e6:.2f}\t{ms:.4f}\t{fl/(ms*1e-3)/1e12 if ms else 0:.2f}\t{by/(ms*1e-3)/1e9 if ms else 0:.1f}\n")
    if rank == 0 and world == 1 and not args.no_other_configs and not e2e and raw_u8 is None and (B, N, S) == (32, 10, 256):
        try:
            result["configs"] = other_configs(model, args.precision, device)
        except Exception as exc:   # noqa: BLE001 -- the extra runs must not cost the headline line
            result["configs"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and raw_u8 is None and (e2e or (N, S) == (10, 256)):
        from oracle import cpu_ref
        base, ref = cpu_baseline_e2e(sd, args.cpu_seconds, Hh, Ww) if e2e else cpu_baseline(sd, args.cpu_seconds)
        result["cpu_baseline"] = base
        nb = min(ref[3].shape[0], outs[3].shape[0])
        got = outs[3][:nb].float().cpu()
        result["parity"] = {"rel_l2": float(f"{cpu_ref.rel_l2(got, ref[3][:nb]):.3e}"),
                            "rmse": float(f"{cpu_ref.rmse(got, ref[3][:nb]):.3e}"),
                            "checked": f"pred3 of the first {nb} stacks of the timed batch vs oracle, gate 1e-3"}
        result["gpu_over_cpu"] = round(value / base["value"], 1)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
