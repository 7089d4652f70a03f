#!/usr/bin/env python3
"""Benchmark of the point-landmark hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch of synthetic input held
in HBM: `--instances` independent filter instances per GPU, each a
point-landmark update of L=150 landmarks (state dim n=462): log-likelihood +
gradient + exact Hessian in fp64, one batched kernel launch.  A *landmark
update* = evaluating (l_j, grad l_j, hess l_j) for one landmark and accumulating
it into the dense g/H; value = landmark updates per second over all ranks.
With --gpus N > 1 (launched by torch.distributed.run) every rank owns its own
instances (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      dominant kernel (point_loglik) against the HBM roof
  cpu_baseline  the CPU oracle timed on this host (N=1 only), reference-faithful
                autodiff pass structure, single thread
  corner        the front-end leg: 1080p frames/s with its own roofline
  dense         (N=1) the 6012 x 6012 square-root QR against the fp64 matrix-core peak and the
                wall time of one complete 150-landmark measurement update
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_LANDMARKS = 150
N_STATE = 12 + 3 * L_LANDMARKS
SIGMA = 7.0
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec


def loglik_algorithmic_bytes(n, L):
    """Bytes one call must move: dense H write + g write + x read + per-landmark
    (y 16 B + visible 1 B) + camera struct/sigma/value (168 B)."""
    return 8 * n * n + 8 * n + 8 * n + 17 * L + 168


def corner_algorithmic_bytes(w, h, k):
    """One u8 frame read + 12-byte keypoint records written."""
    return w * h + 12 * k


def build_scenes(count, seed0):
    from visualslam_amd import synthetic as syn
    cam = syn.read_camera_xml()
    scs = [syn.Scene(L_LANDMARKS, seed0 + i, cam=cam, dense_S=False) for i in range(count)]
    return (np.stack([s.x for s in scs]), np.stack([s.y for s in scs]), np.stack([s.visible for s in scs]))


def cpu_baseline(x, y, vis, budget_s=12.0):
    """Times the CPU oracle's reference-faithful path (n(n+1)/2 second-order dual
    passes per Hessian, src/model.cpp:427-437) on a bounded sample of the same
    workload and extrapolates to whole Hessians."""
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    c = syn.read_camera_xml()
    cam = orc.make_camera(c["K"][0, 0], c["K"][1, 1], c["K"][0, 2], c["K"][1, 2], c["dist"], c["flag"],
                          c["width"], c["height"])
    n = x.shape[1]
    full_passes = n * (n + 1) // 2
    passes, dt, inst = 0, 0.0, 0
    while dt < budget_s:                      # whole Hessians (n(n+1)/2 passes each) until the budget is spent
        t0 = time.perf_counter()
        _, _, _, p = orc.point_loglik_autodiff(x[inst % x.shape[0]], y[inst % x.shape[0]], vis[inst % x.shape[0]], cam, SIGMA)
        dt += time.perf_counter() - t0
        passes += p
        inst += 1
    sec_per_hessian = dt / passes * full_passes
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 1.0:
        orc.point_loglik_analytic(x[reps % x.shape[0]], y[reps % x.shape[0]], vis[reps % x.shape[0]], cam, SIGMA)
        reps += 1
    analytic = reps * L_LANDMARKS / (time.perf_counter() - t0)
    cores = usable_cores()
    allc = all_core_cpu_rate("loglik_analytic", 4.0, cores)
    return {
        "value": L_LANDMARKS / sec_per_hessian, "unit": "landmark updates/s", "cores": 1, "kind": "port",
        "sample": "oracle autodiff-faithful path: %d whole cost evaluations (value+gradient+Hessian, %d dual2nd "
                  "likelihood passes each) of the same C3 instances, %.1f s" % (inst, full_passes, dt),
        "seconds_per_cost_evaluation": sec_per_hessian,
        "analytic_port_updates_per_s": analytic,
        "analytic_port_all_cores": None if allc is None else {
            "value": allc["value"], "unit": "landmark updates/s", "cores": allc["cores"], "kind": "port",
            "sample": "%d worker processes, each %d closed-form cost evaluations of its own 150-landmark instances in %.0f s"
                      % (allc["cores"], allc["units"] // L_LANDMARKS // allc["cores"], allc["seconds_each"])},
        "reference_faithful_all_cores_estimate": L_LANDMARKS / sec_per_hessian * cores,
        "reference_faithful_all_cores_note": "the autodiff-faithful rate above times %d cores (independent instances share nothing); not run" % cores,
    }


def usable_cores(cap=16):
    """Host cores this process may use, at most `cap` (a one-GPU box gives one process a share of 16)."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    return max(1, min(cap, c))


def all_core_cpu_rate(kind, seconds, cores):
    """Units per second of the CPU oracle with one worker process per core, each on its own units (bench_cpu_worker.py);
    the workers are child processes that never touch the GPU."""
    import subprocess
    worker = os.path.join(ROOT, "bench_cpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, kind, str(seconds), str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(cores)]
    rate, units = 0.0, 0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 4 + 180)
        line = [ln for ln in out.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return None
        r = json.loads(line[-1])
        rate += r["units"] / r["seconds"]
        units += r["units"]
    return {"value": rate, "cores": cores, "units": units, "seconds_each": seconds}


def host_info():
    """CPU model and core counts of the node the CPU baselines were timed on."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = None
    return {"cpu_model": model, "nproc": os.cpu_count(), "usable_cores": usable}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--instances", type=int, default=512, help="filter instances per GPU per step")
    ap.add_argument("--frames", type=int, default=64, help="1080p frames per GPU per step (corner leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-corner", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from visualslam_amd import api, synthetic as syn

    from visualslam_amd import sharding
    lay = sharding.RankLayout.from_env()
    rank, local_rank, world = lay.rank, lay.local_rank, lay.world
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; libvslam_hip has no CPU path")
    # Rehearsal of the multi-rank code path on a box with fewer GPUs than ranks: VSLAM_BENCH_SHARE_GPU=1 puts every
    # rank on GPU 0 and VSLAM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).  The numbers
    # of such a run mean nothing; the line says so in "rehearsal".
    share_gpu = os.environ.get("VSLAM_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("VSLAM_BENCH_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B, n, L = args.instances, N_STATE, L_LANDMARKS
    uniq = min(B, 32)
    xs, ys, vs = build_scenes(uniq, sharding.instance_seed(9000, rank, 0))
    reps = (B + uniq - 1) // uniq
    rng = np.random.default_rng(77 + rank)
    x_np = np.tile(xs, (reps, 1))[:B] + 1e-3 * rng.standard_normal((B, n))   # distinct instances
    x = torch.tensor(x_np, device=dev)
    y = torch.tensor(np.tile(ys, (reps, 1))[:B], device=dev)
    vis = torch.tensor(np.tile(vs, (reps, 1))[:B], device=dev)
    val = torch.empty(B, dtype=torch.float64, device=dev)
    g = torch.empty((B, n), dtype=torch.float64, device=dev)
    H = torch.empty((B, n, n), dtype=torch.float64, device=dev)
    code = torch.empty((B, L), dtype=torch.int32, device=dev)

    stream = torch.cuda.current_stream()
    ctx = api.Context(dev_index, stream=stream.cuda_stream)
    cam = api.camera_from_xml(syn.CAMERA_XML)

    def step():
        api.point_loglik_batch_dev(ctx, cam, n, x, y, vis, SIGMA, val, g, H, code)

    for _ in range(args.warmup):
        step()
    barrier()
    # one pair of HIP events around the K steps, on the kernels' stream: an event record between two steps is a packet
    # of its own in the queue and costs the stream about 5 us each, 6 % of a step of this length
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(stream)
    for i in range(args.steps):
        step()
    ev_b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = ev_a.elapsed_time(ev_b) / args.steps
    elapsed = sharding.max_over_ranks(elapsed, device=dev)
    # spread of single steps: 30 further steps after the timed region, each between its own pair of events (these
    # carry the cost of the event records and do not enter `value` or the roofline figures)
    extra = 30
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(extra)]
    for a, b in ev2:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    step_ms = [a.elapsed_time(b) for a, b in ev2]
    q1, med, q3 = (float(v) for v in np.percentile(step_ms, [25, 50, 75]))
    # results to rank 0 (outside the timed region): the only exchange a sharded batch needs
    all_vals = sharding.gather_to_root(val)
    gathered_ok = None if all_vals is None else bool(all(torch.isfinite(v).all().item() for v in all_vals) and len(all_vals) == world)
    # ... with the payload a bank of filters hands over: posterior mean + packed upper-triangular factor per filter
    # (a small bank per rank runs one lock-step measurement update here, outside the timed region)
    filt_msg = None
    nb_f = 16
    f_mo = torch.zeros((nb_f, n), dtype=torch.float64, device=dev)
    f_So = torch.zeros((nb_f, n, n), dtype=torch.float64, device=dev)
    f_err, f_npd = None, -1
    try:
        scs = [syn.Scene(L, sharding.instance_seed(9500, rank, i), dense_S=True) for i in range(4)]
        f_mu = torch.tensor(np.stack([scs[i % 4].x for i in range(nb_f)]), device=dev)
        f_S = torch.tensor(np.stack([np.ascontiguousarray(scs[i % 4].S.T) for i in range(nb_f)]), device=dev)
        f_y = torch.tensor(np.stack([scs[i % 4].y for i in range(nb_f)]), device=dev)
        f_v = torch.tensor(np.stack([scs[i % 4].visible for i in range(nb_f)]), device=dev)
        _, _, f_status = api.measurement_update_iekf_batch_dev(ctx, cam, f_mu, f_S, f_y, f_v, SIGMA, f_mo, f_So)
        f_npd = int(f_status.sum())
        del f_mu, f_S, f_y, f_v
    except Exception as exc:      # the headline measurement above must not be lost to a failure here
        f_err = repr(exc)
    # every rank takes part in the gather whatever happened to its own update: a rank that skipped the collective
    # would leave the others waiting
    t_g = time.perf_counter()
    msg = sharding.gather_filter_results_to_root(f_mo, f_So)
    torch.cuda.synchronize()
    t_g = time.perf_counter() - t_g
    if msg is not None:
        mu_all, S_all = sharding.unpack_filter_results(msg, n)
        filt_msg = {"filters_per_rank": nb_f, "doubles_per_filter": int(msg.shape[1]), "bytes_per_rank": int(msg.shape[1]) * 8 * nb_f,
                    "gather_ms": t_g * 1e3, "filters_on_root": int(msg.shape[0]),
                    "own_block_round_trip_exact": bool(torch.equal(mu_all[:nb_f], f_mo) and torch.equal(S_all[:nb_f], torch.tril(f_So))),
                    "not_positive_definite_on_rank0": f_npd}
        if f_err:
            filt_msg["error_on_rank0"] = f_err
    del f_mo, f_So, msg

    # the two kernels of the call apart (outside the timed region): events inside the library, on its stream
    ctx.set_kernel_timing(True)
    split = []
    for _ in range(20):
        step()
        split.append(ctx.last_point_loglik_times())
    ctx.set_kernel_timing(False)
    terms_ms, assemble_ms = (float(np.mean([s_[i] for s_ in split])) for i in (0, 1))

    # single-instance latency of the same update (launch-latency bound; reported, not the headline)
    x1, y1, v1 = x[:1], y[:1], vis[:1]
    for _ in range(5):
        api.point_loglik_batch_dev(ctx, cam, n, x1, y1, v1, SIGMA, val[:1], g[:1], H[:1], code[:1])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(200):
        api.point_loglik_batch_dev(ctx, cam, n, x1, y1, v1, SIGMA, val[:1], g[:1], H[:1], code[:1])
    e1.record(stream)
    torch.cuda.synchronize()
    single_us = e0.elapsed_time(e1) / 200 * 1e3

    # the same call on the whole 4 096-instance batch of BASELINE configs[4] held by one GPU (reported beside the
    # headline, never in place of it: the headline keeps 512 instances per GPU, configs[4]'s share of one of 8 GPUs)
    whole_batch = None
    if rank == 0 and world == 1 and B < 4096:
        B4 = 4096
        r4 = (B4 + uniq - 1) // uniq
        x4 = torch.tensor(np.tile(xs, (r4, 1))[:B4] + 1e-3 * rng.standard_normal((B4, n)), device=dev)
        y4 = torch.tensor(np.tile(ys, (r4, 1))[:B4], device=dev)
        v4 = torch.tensor(np.tile(vs, (r4, 1))[:B4], device=dev)
        val4 = torch.empty(B4, dtype=torch.float64, device=dev)
        g4 = torch.empty((B4, n), dtype=torch.float64, device=dev)
        H4 = torch.empty((B4, n, n), dtype=torch.float64, device=dev)
        code4 = torch.empty((B4, L), dtype=torch.int32, device=dev)
        for _ in range(3):
            api.point_loglik_batch_dev(ctx, cam, n, x4, y4, v4, SIGMA, val4, g4, H4, code4)
        e4a, e4b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e4a.record(stream)
        for _ in range(30):
            api.point_loglik_batch_dev(ctx, cam, n, x4, y4, v4, SIGMA, val4, g4, H4, code4)
        e4b.record(stream)
        torch.cuda.synchronize()
        ms4 = e4a.elapsed_time(e4b) / 30
        whole_batch = {"instances": B4, "ms_per_call": ms4, "landmark_updates_per_s": B4 * L / (ms4 * 1e-3),
                       "frac_of_hbm_roof": loglik_algorithmic_bytes(n, L) * B4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "all_finite": bool(torch.isfinite(val4).all().item()),
                       "note": "30 calls between one pair of HIP events, after the timed region; same kernels, 8x the instances per launch"}
        del x4, y4, v4, val4, g4, H4, code4

    corner = None
    if not args.no_corner:
        try:
            from bench_corner import run_corner_leg
            corner = run_corner_leg(args, ctx, dev, stream, rank, world, barrier)
        except ImportError:
            corner = None

    dense = None
    if rank == 0 and world == 1 and not args.no_dense:
        from bench_dense import run_dense_leg
        dense = run_dense_leg(ctx, dev, stream)

    if rank == 0:
        units = B * L * args.steps * world
        alg = loglik_algorithmic_bytes(n, L) * B
        achieved = alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch: not measured in this run (counters need rocprofv3 around the process) but replayed from
        # the committed counter passes -- and only when this run's assemble kernel lasts what it lasted under those
        # passes (within 5 %), i.e. when the profile belongs to the kernels that ran here
        traffic, traffic_source = None, "no committed counter pass for this batch size"
        tfile = os.path.join(ROOT, "profiles", "traffic_point_loglik.json")
        if os.path.exists(tfile) and B == 512:
            tj = json.load(open(tfile))
            ref_ms = tj.get("assemble_kernel_avg_ms")
            if ref_ms is None:
                traffic_source = "profiles/traffic_point_loglik.json carries no kernel duration to check this run against: not replayed"
            elif abs(assemble_ms - ref_ms) <= 0.05 * ref_ms:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("replayed from profiles/traffic_point_loglik.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "tools/run_loglik.py 512 150, FETCH doubled per the gfx950 note); checked in this run: loglik_assemble_kernel "
                                  "%.4f ms here against %.4f ms in the profiled run (within 5 %%)" % (assemble_ms, ref_ms))
            else:
                traffic_source = ("NOT replayed: loglik_assemble_kernel lasts %.4f ms here against %.4f ms in the run "
                                  "profiles/traffic_point_loglik.json was taken from" % (assemble_ms, ref_ms))
        out = {
            "metric": "landmark Jacobian updates/s", "value": units / elapsed, "unit": "landmark updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "step_stats": {"samples": len(step_ms), "median_ms": med, "q1_ms": q1, "q3_ms": q3, "iqr_ms": q3 - q1,
                           "median_frac_of_hbm_roof": loglik_algorithmic_bytes(n, L) * B / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "%d steps after the timed region, each between its own pair of HIP events on the kernels' "
                                   "stream; the event records add about 10 us to a step timed this way, which the timed "
                                   "region (one pair of events around all K steps) does not pay" % extra},
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3 x%d: %d independent point-landmark updates per GPU per step, L=150 landmarks, "
                                   "n=462 states, log-lik + gradient + exact Hessian (BASELINE configs[2] batched as configs[4])"
                                   % (B, B),
                       "instances_per_gpu": B, "landmarks": L, "state_dim": n, "sigma_px": SIGMA,
                       "parallelism": "replicated instances, no collective"},
            "roofline": {"bound": "hbm", "kernel": "loglik_terms_kernel + loglik_assemble_kernel (one point_loglik call)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": kern_ms,
                         "kernels_apart": {"loglik_terms_kernel_ms": terms_ms, "loglik_assemble_kernel_ms": assemble_ms,
                                           "assemble_alone_frac": alg / (assemble_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "events recorded by the library between its two launches; achieved/frac "
                                                   "above price the whole call (both kernels and the gap between them)"}},
            "results_gathered_to_root": gathered_ok,
            "filter_results_gathered_to_root": filt_msg,
            "single_instance_latency_us": single_us,
            "single_instance_updates_per_s": L / (single_us * 1e-6),
            "host": host_info(),
        }
        if share_gpu or backend != "nccl":
            out["rehearsal"] = "ranks share GPU 0 = %s, backend %s: exercises the multi-rank code path only, the numbers are not a measurement" % (share_gpu, backend)
        if whole_batch is not None:
            out["whole_configs4_batch_on_one_gpu"] = whole_batch
        if corner is not None:
            out["corner"] = corner
        if dense is not None:
            out["dense"] = dense
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x_np, np.tile(ys, (reps, 1))[:B], np.tile(vs, (reps, 1))[:B])
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
