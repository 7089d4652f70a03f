"""Corner front-end leg of bench.py: 1080p frames/s (Harris + Shi-Tomasi keypoint
lists per frame: response -> 3x3 NMS -> threshold -> score-sorted top-K)."""
import json
import os
import time

import numpy as np

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0


def issue_roofline(counter_file, kernel_part, step_ms, launches_per_step=1):
    """The front-end kernels are bound by instruction issue, not by HBM: a wave64 VALU instruction occupies one of the
    1 024 SIMDs (256 compute units x 4) for four cycles.  From the committed SQ counter pass (SQ_INSTS_VALU, mean per
    launch) this gives the time the kernel's instruction stream needs at best, to set beside the measured step."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", counter_file)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        per_kernel = json.load(fh).get("mean_per_launch", {})
    hit = [v for k, v in per_kernel.items() if kernel_part in k and "SQ_INSTS_VALU" in v]
    if not hit:
        return None
    insts = float(hit[0]["SQ_INSTS_VALU"]) * launches_per_step
    simds, cycles_per_inst, clock_hz = 1024, 4, 2.4e9
    bound_ms = insts * cycles_per_inst / simds / clock_hz * 1e3
    return {"bound": "valu issue", "kernel": kernel_part, "wave_instructions_per_step": insts, "peak": "1024 SIMDs x 1 instruction / 4 cycles at 2.4 GHz",
            "issue_time_ms": bound_ms, "step_ms": step_ms, "frac": bound_ms / step_ms,
            "source": "SQ_INSTS_VALU replayed from profiles/%s (rocprofv3 --pmc pass); frac = issue time / measured step (sort, merge and emit kernels included in the step)" % counter_file}


def run_corner_leg(args, ctx, dev, stream, rank, world, barrier):
    import torch
    import torch.distributed as dist
    from visualslam_amd import api, synthetic as syn

    F = args.frames
    base = np.stack([syn.frame(5000 + 8 * rank + i) for i in range(min(F, 8))])
    frames = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    tb = torch.tensor(base, device=dev)
    for i in range(F):
        frames[i] = torch.roll(tb[i % tb.shape[0]], shifts=(i // tb.shape[0]) * 7, dims=1)   # distinct frames
    params = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=10000)
    cap = 10000
    out = torch.zeros((F, 2, cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros((F, 2), dtype=torch.int32, device=dev)

    def step():
        api.corner_detect_batch_dev(ctx, frames, W, H, W, W * H, F, params, api_both, out, cap, cnt)

    api_both = 2
    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # keypoint lists to rank 0 (outside the timed region): counts, then the padded records (Harris lists)
    kp_msg = None
    try:
        from visualslam_amd import sharding
        t_g = time.perf_counter()
        got = sharding.gather_keypoints_to_root(cnt[:, 0].contiguous(), out[:, 0].contiguous())
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t_g
        if got is not None:
            call, rall = got
            k0 = int(call[0])
            kp_msg = {"frames_on_root": int(call.shape[0]), "records_per_frame": int(rall.shape[1]), "bytes_per_rank": int(rall.shape[1]) * 12 * F + 4 * F,
                      "gather_ms": t_g * 1e3, "own_block_round_trip_exact": bool(torch.equal(rall[0, :k0], out[0, 0, :k0]))}
    except Exception as exc:
        kp_msg = {"error": repr(exc)}
    if rank != 0:
        return None
    counts = cnt.cpu().numpy()
    assert (counts >= 0).all(), "candidate list overflow"
    kp = float(counts.sum()) / F
    alg = (W * H + 12.0 * kp) * F
    achieved = alg / (dev_ms * 1e-3) / 1e9
    traffic = None                      # HBM bytes per step from the committed counter passes (64-frame step)
    tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_corner.json")
    if F == 64 and os.path.exists(tfile):
        with open(tfile) as fh:
            traffic = json.load(fh).get("hbm_bytes_per_step")
    res = {
        "metric": "corner-detect frames/s (1080p)", "value": F * args.steps * world / elapsed, "unit": "frames/s",
        "frames_per_gpu_per_step": F, "ms_per_step": elapsed / args.steps * 1e3,
        "mean_keypoints_per_frame": {"harris": float(counts[:, 0].mean()), "shi_tomasi": float(counts[:, 1].mean())},
        "roofline": {"bound": "hbm", "kernel": "corner_strip_kernel<3> + sort/merge/emit", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "replayed from profiles/traffic_corner.json (rocprofv3 --pmc passes of tools/run_corner.py 64); not re-measured in this run",
                     "algorithmic_bytes_per_step": alg, "device_ms_per_step": dev_ms},
        "dtype": "u8 -> int32 structure tensor -> f32 response",
        "keypoints_gathered_to_root": kp_msg,
    }
    if F == 64:
        res["issue_roofline"] = issue_roofline("r03_corner_sq_counters.json", "corner_strip_kernel", dev_ms)
    # FAST-9/16 (first stage of cv::ORB::detect) on the same frames: threshold 20 as the reference configures it
    fout = torch.zeros((F, cap, 3), dtype=torch.int32, device=dev)
    fcnt = torch.zeros(F, dtype=torch.int32, device=dev)
    for _ in range(3):
        api.fast9_detect_batch_dev(ctx, frames, W, H, W, W * H, F, 20, True, cap, fout, cap, fcnt)
    torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    freps = 20
    f0.record(stream)
    for _ in range(freps):
        api.fast9_detect_batch_dev(ctx, frames, W, H, W, W * H, F, 20, True, cap, fout, cap, fcnt)
    f1.record(stream)
    torch.cuda.synchronize()
    fast_ms = f0.elapsed_time(f1) / freps
    fk = float(fcnt.clamp(min=0).sum().item()) / F
    falg = (W * H + 12.0 * fk) * F
    ftraffic = None
    ffile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_fast9.json")
    if F == 64 and os.path.exists(ffile):
        with open(ffile) as fh:
            ftraffic = json.load(fh).get("hbm_bytes_per_step")
    res["fast9"] = {"value": F / (fast_ms * 1e-3), "unit": "frames/s", "ms_per_step": fast_ms, "threshold": 20, "nonmax": True,
                    "mean_keypoints_per_frame": fk, "overflowed_frames": int((fcnt < 0).sum().item()),
                    "roofline": {"bound": "hbm", "kernel": "fast9_tile_kernel + sort/merge/emit", "achieved": falg / (fast_ms * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": falg / (fast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": ftraffic,
                                 "traffic_source": "replayed from profiles/traffic_fast9.json (rocprofv3 --pmc passes of tools/run_fast9.py 64); not re-measured in this run",
                                 "algorithmic_bytes_per_step": falg}}
    if F == 64:
        res["fast9"]["issue_roofline"] = issue_roofline("r03_fast9_sq_counters.json", "fast9_tile_kernel", fast_ms)
    # ORB as the reference configures it: level-0 keypoints (FAST -> edge -> 2n by FAST score -> Harris at the corners
    # -> n best) for all frames in one call, then blur + orientation + descriptors frame by frame; and the whole
    # 8-level pyramid of one host frame through the host-pointer entry (PCIe and the level loop included)
    try:
        n0 = int(api.orb_level_plan(api.orb_params(), W, H)[2][0])
        ocap = 16384
        oout = torch.zeros((F, ocap, 3), dtype=torch.int32, device=dev)
        ocnt = torch.zeros(F, dtype=torch.int32, device=dev)
        oblur = torch.zeros((H, W), dtype=torch.uint8, device=dev)
        oang = torch.zeros((F, ocap), dtype=torch.float32, device=dev)
        odesc = torch.zeros((F, ocap, 32), dtype=torch.uint8, device=dev)
        opat = torch.tensor(api.orb_default_pattern(), device=dev)

        def orb_detect_step():
            api.orb_detect_level_batch_dev(ctx, frames, W, H, W, W * H, F, n0, oout, ocap, ocnt)

        def orb_describe_step():
            for i in range(F):
                api.gaussian_blur7_dev(ctx, frames[i], oblur)
                api.orb_describe_level_dev(ctx, frames[i], oblur, oout[i], ocnt[i:i + 1], n0 + 64, opat, oang[i], odesc[i])

        orb = {}
        for name, fn, reps in (("detect_level0", orb_detect_step, 10), ("blur_orient_describe_level0", orb_describe_step, 3)):
            fn()
            torch.cuda.synchronize()
            f0.record(stream)
            for _ in range(reps):
                fn()
            f1.record(stream)
            torch.cuda.synchronize()
            ms = f0.elapsed_time(f1) / reps
            orb[name] = {"ms_per_step": ms, "frames_per_s": F / (ms * 1e-3)}
        orb["mean_keypoints_level0"] = float(ocnt.clamp(min=0).float().mean().item())
        # the whole front end, all eight levels, keypoints + angles + descriptors, one call for the batch
        bcap = 12288
        bout = torch.zeros((F, bcap, 4), dtype=torch.int32, device=dev)
        bcnt = torch.zeros(F, dtype=torch.int32, device=dev)
        bang = torch.zeros((F, bcap), dtype=torch.float32, device=dev)
        bdesc = torch.zeros((F, bcap, 32), dtype=torch.uint8, device=dev)
        for name, kw in (("all_levels_detect", {}), ("all_levels_detect_and_describe", {"angle": bang, "desc": bdesc})):
            api.orb_detect_and_compute_batch_dev(ctx, frames, bout, bcnt, **kw)
            torch.cuda.synchronize()
            f0.record(stream)
            for _ in range(3):
                api.orb_detect_and_compute_batch_dev(ctx, frames, bout, bcnt, **kw)
            f1.record(stream)
            torch.cuda.synchronize()
            ms = f0.elapsed_time(f1) / 3
            orb[name] = {"ms_per_step": ms, "frames_per_s": F / (ms * 1e-3)}
        orb["mean_keypoints_all_levels"] = float(bcnt.clamp(min=0).float().mean().item())
        orb["frames_over_capacity"] = int((bcnt < 0).sum().item())
        del bout, bcnt, bang, bdesc
        hctx = api.Context(dev.index or 0)
        api.orb_detect_and_compute(hctx, base[0])
        t0 = time.perf_counter()
        kpo, ango, desco, levo = api.orb_detect_and_compute(hctx, base[0])
        orb["host_frame_8_levels_ms"] = (time.perf_counter() - t0) * 1e3
        orb["host_frame_keypoints"] = int(kpo.size)
        hctx.close()
        res["orb"] = orb
        del oout, ocnt, oblur, oang, odesc
    except Exception as exc:                # reported, not fatal: the headline metric above stands on its own
        res["orb"] = {"error": repr(exc)}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            orc.fast9_detect(base[reps % len(base)], 20, True)
            reps += 1
        res["fast9"]["cpu_baseline"] = {"value": reps / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "port",
                                        "sample": "%d synthetic 1080p frames, oracle scalar C" % reps}
        op = orc.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=10000)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 5.0:
            orc.corner_detect(base[reps % len(base)], op, 0)
            orc.corner_detect(base[reps % len(base)], op, 1)
            reps += 1
        res["cpu_baseline"] = {"value": reps / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d synthetic 1080p frames, both detectors, oracle scalar C" % reps}
        from bench import all_core_cpu_rate, usable_cores
        allc = all_core_cpu_rate("corner", 4.0, usable_cores())
        if allc is not None:
            res["cpu_baseline_all_cores"] = {"value": allc["value"], "unit": "frames/s", "cores": allc["cores"], "kind": "port",
                                             "sample": "%d worker processes, frame-level parallelism, %d synthetic 1080p frames in %.0f s each, both detectors"
                                                       % (allc["cores"], allc["units"], allc["seconds_each"])}
    return res
