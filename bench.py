#!/usr/bin/env python3
"""Benchmark of the ICP / Go-ICP correspondence hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (N = 1): the exact squared Euclidean distance transform of the bunny045 scan on a
300^3 grid (27 M voxels), voxelised by the reference's box rule; one step = one full build (clear,
splat, census, z+y pass, x pass) with the target cloud already resident in HBM. `value` is voxels/s.
For N > 1 the path does not shard (every pass needs whole rows along its axis): every rank builds its
own replica and `value` is the aggregate over ranks (weak scaling, no data-path collective).

The same JSON line carries
  roofline      the dominant EDT kernel (pass x) priced against the 8 TB/s HBM peak from its
                algorithmic bytes (8 B per voxel: one int32 read + one int32 write), timed with HIP
                events on 40 builds of their own right after the timed region (so the figure does not
                depend on --steps);
  roofline_nn   one ICP pass over the synthetic 1 M x 1 M pair (36 B per query + 12 B per target point),
                timed with events on the stream the kernels run on: the first pass (queries 10 degrees
                off, far from the surface) and the converged passes separately;
  cpu_baseline  the CPU restatement of the reference's three-pass EDT (oracle/, OpenMP) on this box's
                host cores, on a bounded number of repetitions of the same 300^3 build, at all cores and
                at the 2 threads the reference hard-codes (src/SingleThreadIcp.cpp:104);
  extra         NN / ICP-iteration throughput of the synthetic 1 M x 1 M cloud pair (query-sharded over
                the ranks, one 16-double all-reduce per iteration), the 16 M x 16 M pair, DT-lookup
                throughput and Go-ICP time-to-optimum on the reference's bunny pair.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (spec)
DIV = 300


def load_scan(name):
    return np.fromfile(os.path.join(ROOT, "tests", "golden", name + ".xyz.f32"), dtype="<f4").reshape(-1, 3).astype(np.float32)


def synth_surface(n, seed, noise=0.0):
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, n)
    phi = rng.uniform(0.0, 2 * np.pi, n)
    theta = np.arccos(u)
    r = 0.35 * (1 + 0.15 * np.sin(3 * theta) * np.cos(5 * phi))
    p = np.stack([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)], 1)
    if noise:
        p = p + rng.normal(0.0, noise, p.shape)
    return p.astype(np.float32)


def rigid(axis, deg, t):
    a = np.asarray(axis, float); a /= np.linalg.norm(a)
    th = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def _newest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except Exception:
        return None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest rocprofv3 counter summary under profiles/
    (2 * FETCH_SIZE + WRITE_SIZE, KiB, collected in separate --pmc passes by tools/prof_round.sh); None when
    no summary is present. Offline evidence, not measured by this run. On gfx950 FETCH_SIZE reports half of the
    bytes read (128-byte requests tallied at 64), for 4-byte as for 16-byte loads per lane, and WRITE_SIZE the
    bytes written exactly: tools/fetch_calib.hip streams a known 1 GiB in both widths and reads 0.500 / 1.000."""
    data = _newest_profile("r*_edt_pmc.json")
    if not data:
        return None
    for name, c in data.items():
        if isinstance(c, dict) and name.startswith(kernel) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    return None


def measured_nn_traffic():
    """HBM bytes of one converged ICP pass over the 1 M x 1 M pair, summed over its kernels, from the newest counter
    summary profiles/r*_icp_pmc.json (tools/pmc_icp1m.sh); None when there is none. Offline evidence."""
    data = _newest_profile("r*_icp_pmc.json")
    if not data:
        return None
    try:
        return sum((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 for c in data["converged_pass"].values())   # see measured_traffic
    except Exception:
        return None


def measured_nn_valu_issue():
    """Vector instructions issued x 4 cycles over the cycles of all 1 024 SIMDs, summed over the kernels of one ICP pass of the
    1 M x 1 M pair (first pass, converged pass), from the newest counter summary under profiles/ (SQ_INSTS_VALU and
    GRBM_GUI_ACTIVE, the latter summed over the 8 XCDs); None where the summary lacks the counters. Offline evidence."""
    data = _newest_profile("r*_icp_pmc.json")
    out = {}
    for which in ("first_pass", "converged_pass"):
        try:
            ks = data[which].values()
            out[which] = sum(c["SQ_INSTS_VALU"] for c in ks) * 4.0 / 1024.0 / sum(c["GRBM_GUI_ACTIVE"] / 8.0 for c in ks)
        except Exception:
            out[which] = None
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-extra", action="store_true", help="skip the NN / ICP / lookup extras")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--nn-points", type=int, default=1_000_000)
    ap.add_argument("--big-points", type=int, default=16_000_000, help="points of the large scan-to-model pair (0 = skip)")
    ap.add_argument("--no-goicp", action="store_true", help="skip the Go-ICP bunny-270 extra")
    ap.add_argument("--goicp-lookahead", type=int, default=1, help="queued translation cubes whose children are evaluated ahead of time")
    ap.add_argument("--goicp-speculate", type=int, default=255, help="rotation cubes whose inner searches run ahead of time in the same launches")
    ap.add_argument("--goicp-host-search", action="store_true", help="drive the inner searches from the host (a round trip per step) instead of running them on the device")
    ap.add_argument("--goicp-share-launches", action="store_true", help="several ranks: cut every bound batch into shares (a collective per launch) instead of dealing out the rotation cubes")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (gloo for CPU-side rehearsal)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--extra-limit", type=float, default=420.0,
                    help="seconds the legs after the headline may take before the line is printed without the rest")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from icp_amd.api import Context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx = Context(local)
    tgt = load_scan("bunny045")
    tgt_dev = torch.from_numpy(tgt).to(dev)
    ctx.set_target(tgt_dev)            # target resident in HBM before anything is timed
    ctx.edt_build(DIV)
    info = ctx.edt_info()
    voxels = DIV ** 3

    # ---- headline: EDT build ------------------------------------------------------------------------
    for _ in range(args.warmup):
        ctx.edt_rebuild_async()
    ctx.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):            # K back-to-back builds, no events: the timed region
        ctx.edt_rebuild_async()
    ctx.sync()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    value = world * voxels * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    # per-kernel times: 40 further builds, every launch between its own start / stop HIP events on the library's
    # stream (a dispatch with events costs about 5 us of stream time more than one without, so these stay out of `value`)
    roofline_samples = 40
    stage_ms = ctx.edt_rebuild_timed(roofline_samples, sample_every=1)
    barrier()

    stage_names = ["splat", "census", "pass_zy", "pass_x"]
    per_stage = {k: float(v) for k, v in zip(stage_names, stage_ms)}
    dom = "pass_x"
    alg_bytes = 8.0 * voxels                       # one int32 read + one int32 write per voxel
    achieved = alg_bytes / (per_stage[dom] * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "k_edt_pass_x", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic("k_edt_pass_x"),
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_stage[dom], "samples": roofline_samples,
                # The whole step priced with the bytes this design moves by construction: the occupancy bits in (1/8 B per
                # voxel), the in-plane distances out of pass zy (4 B) and through pass x (4 B in, 4 B out). An upper bound on
                # what it moves in fact: planes and rows without voxels are skipped (compare measured_traffic_bytes).
                "whole_build": {"algorithmic_bytes": 12.125 * voxels, "GB/s": 12.125 * voxels / (ms_per_step * 1e-3) / 1e9,
                                "frac": 12.125 * voxels / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "kernel_ms_sum": sum(per_stage.values()),
                                "measured_traffic_bytes": sum(filter(None, (measured_traffic(k) for k in ("k_edt_splat", "k_edt_census", "k_edt_pass_zy", "k_edt_pass_x")))) or None},
                "stage_ms": per_stage}

    # The first build for a target is the one the reference's use pays for (one LDT per target, src/testLDT.cpp:40-212):
    # a fresh context, the target cloud handed over again, one checked build with start / stop events on every kernel.
    try:
        cold = []
        for _ in range(3):
            c0 = Context(local)
            c0.set_target(tgt_dev)
            c0.edt_set_timing(True)
            t0 = time.perf_counter(); c0.edt_build(DIV); wall = time.perf_counter() - t0
            ms, attempts = c0.edt_last_build()
            cold.append((float(ms.sum()), [float(v) for v in ms], attempts, wall))
            c0.close()
        best = min(cold)
        roofline["first_build"] = {"stage_ms": dict(zip(stage_names, best[1])), "kernel_ms_sum": best[0], "attempts": best[2],
                                   "voxels_per_s": voxels / (best[0] * 1e-3), "pass_x_frac": alg_bytes / (best[1][3] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "call_wall_ms": best[3] * 1e3,
                                   "note": "best of three fresh contexts; call_wall_ms also holds the allocation of the grids (0.33 GB) and the read-back of the build's counters"}
    except Exception as exc:
        roofline["first_build"] = {"error": repr(exc)[:200]}

    extra = {}
    roofline_nn = None
    roofline_lookup = None
    roofline_bnb = None
    cpu = None

    # The legs below share collectives between the ranks: a rank that fails inside one would leave the others waiting
    # in theirs. The headline is measured by now, so a watchdog prints the line with what there is when the legs
    # overrun their limit, and ends the process (every rank has one; exactly one line is printed either way).
    import threading
    emit_lock = threading.Lock()
    emitted = [False]

    def snapshot(obj):
        # the main thread may be adding keys while the watchdog thread copies: try again a few times, then give up on the part
        import copy
        for _ in range(20):
            try:
                return copy.deepcopy(obj)
            except RuntimeError:
                time.sleep(0.01)
        return {"errors": {"snapshot": "could not copy a dictionary that was being written"}}

    def emit(timed_out=False):
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            try:
                ex, cp, rn = snapshot(extra), snapshot(cpu), snapshot(roofline_nn)
                rl, rb = snapshot(roofline_lookup), snapshot(roofline_bnb)
                if timed_out:
                    ex.setdefault("errors", {})["timeout"] = f"legs after the headline exceeded {args.extra_limit} s; line printed without the rest"
                if rank == 0:
                    line = {"metric": "EDT voxels/s", "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
                            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                            "vs_baseline": None, "dtype": "int32", "data": "bunny045 scan (40097 points) voxelised on a 300^3 grid",
                            "config": {"workload": "EDT 300^3 from bunny045 (reference box rule), full build per step",
                                       "div": DIV, "occupied_voxels": int(info["occupied"]), "parallelism": "replicas only" if world > 1 else "1 GPU"},
                            "roofline": roofline, "roofline_nn": rn, "roofline_lookup": rl, "roofline_bnb": rb, "cpu_baseline": cp, "extra": ex}
                    print(json.dumps(line), flush=True)
            finally:
                if timed_out:
                    # a leg hung (a rank stuck in a collective, a GPU call that never returns): the headline is out, but the
                    # run did not finish, and the exit status has to say so
                    sys.stdout.flush()
                    os._exit(3)

    watchdog = threading.Timer(args.extra_limit, emit, kwargs={"timed_out": True})
    watchdog.daemon = True
    watchdog.start()

    if not args.no_extra:
        try:
            # DT lookup: bunny000 points against the grid
            q = torch.from_numpy(load_scan("bunny000")).to(dev)
            qrep = q.repeat(25, 1).contiguous()       # ~1 M queries
            ctx.edt_lookup(qrep)
            barrier(); t0 = time.perf_counter()
            for _ in range(20):
                ctx.edt_lookup(qrep)
            barrier()
            el = max_over_ranks(time.perf_counter() - t0)
            extra["dt_lookup_queries_per_s"] = world * qrep.shape[0] * 20 / el
            # the lookup kernel on a batch large enough that its launch does not count: 16 M queries spread over the grid's
            # box and a little beyond, 20 B per query (12 B in, one 4-byte gather, 4 B out), events on the stream it runs on
            nq = 1 << 24
            gen = torch.Generator(device=dev); gen.manual_seed(7)
            lo_b = torch.tensor(info["min"], device=dev) - 0.02
            span = float(info["cell"]) * DIV + 0.04
            qbig = (torch.rand((nq, 3), device=dev, generator=gen) * span + lo_b).contiguous()
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.edt_lookup(qbig)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            times = []
            for _ in range(10):
                e0.record(); ctx.edt_lookup(qbig); e1.record(); torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ctx.reset_stream()
            lk_ms = float(np.median(times))
            roofline_lookup = {"bound": "hbm", "kernel": "k_edt_lookup", "queries": nq, "algorithmic_bytes_per_launch": 20.0 * nq,
                               "avg_launch_ms": lk_ms, "achieved": 20.0 * nq / (lk_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": 20.0 * nq / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "queries_per_s": nq / (lk_ms * 1e-3),
                               "traffic": None, "note": "uniform queries over the whole box: the gather touches all 108 MB of the grid"}
            del qbig
        except Exception as exc:      # an extra leg must not take the headline line with it
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["dt_lookup"] = repr(exc)[:300]

        try:
            # configs[2]: synthetic 1 M x 1 M ICP, query-sharded
            n = args.nn_points
            tgt_s = synth_surface(n, 42)
            src_all = synth_surface(n, 43, noise=1e-3).astype(np.float64)
            Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
            src_all = (src_all @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
            lo, hi = rank * n // world, (rank + 1) * n // world
            ctx2 = Context(local)
            ctx2.set_target(torch.from_numpy(tgt_s).to(dev))
            ctx2.set_source(torch.from_numpy(np.ascontiguousarray(src_all[lo:hi])).to(dev))
            sums = torch.zeros(16, dtype=torch.float64, device=dev)
            ctx2.set_stream(torch.cuda.current_stream().cuda_stream)
            iters = 30
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pass_ms = []
            from icp_amd.host import kabsch_from_sums

            def icp_iter(T, n_total, c):
                ev0.record()
                c.icp_step_async(T, sums)       # on torch's current stream (set_stream above), so the events bracket its kernels
                ev1.record()
                if world > 1 and args.backend == "nccl":
                    dist.all_reduce(sums)          # RCCL, same stream as the kernels: no host round trip
                    s = sums.cpu().numpy()
                elif world > 1:
                    hs = sums.cpu(); dist.all_reduce(hs); s = hs.numpy()
                else:
                    s = sums.cpu().numpy()
                pass_ms.append(ev0.elapsed_time(ev1))
                return kabsch_from_sums(s, n_total) @ T

            T = np.eye(4)
            barrier(); t0 = time.perf_counter()
            for _ in range(iters):
                T = icp_iter(T, n, ctx2)
            barrier()
            el = max_over_ranks(time.perf_counter() - t0)
            extra["icp_1m_iterations_per_s"] = iters / el
            extra["nn_queries_per_s"] = n * iters / el
            extra["icp_1m_rot_err_deg"] = float(np.rad2deg(np.arccos(np.clip((np.trace(T[:3, :3] @ Tm[:3, :3]) - 1) / 2, -1, 1))))
            extra["nn_roofline_frac"] = (36.0 * n + 12.0 * n * world) * iters / el / 1e9 / (HBM_PEAK_GBS * world)
            extra["icp_1m_pass_ms"] = [round(x, 4) for x in pass_ms]
            if world == 1:
                # the same 30 iterations as one call (SingleThreadIcp::registration's loop on the device, icpx_icp_run: no
                # round trip per iteration, the 3 x 3 problem solved by one GPU thread with the host classes' arithmetic),
                # from the same start: the source is set again so that no pair of the run above is left
                ctx2.set_source(torch.from_numpy(np.ascontiguousarray(src_all[lo:hi])).to(dev))
                mt0 = tgt_s.astype(np.float64).sum(0) / n
                torch.cuda.synchronize(); ctx2.sync()
                t0 = time.perf_counter()
                Td, _, _ = ctx2.icp_run(np.eye(4), iters, mt0)
                el_dev = time.perf_counter() - t0
                extra["icp_1m_iterations_per_s_host_driven"] = extra["icp_1m_iterations_per_s"]
                extra["icp_1m_iterations_per_s"] = iters / el_dev
                extra["nn_queries_per_s"] = n * iters / el_dev
                extra["icp_1m_loop"] = "on the device (icpx_icp_run); icp_1m_pass_ms and roofline_nn are from the host-driven passes"
                extra["icp_1m_device_loop_rot_err_deg"] = float(np.rad2deg(np.arccos(np.clip((np.trace(Td[:3, :3] @ Tm[:3, :3]) - 1) / 2, -1, 1))))
                extra["nn_roofline_frac"] = (36.0 * n + 12.0 * n) * iters / el_dev / 1e9 / HBM_PEAK_GBS
            for _ in range(10):                   # ten more passes, converged by now: the steady state of the pass
                T = icp_iter(T, n, ctx2)
            conv_ms = float(np.mean(pass_ms[-10:]))
            alg = 36.0 * (hi - lo) + 12.0 * n      # this rank's queries + the whole target, bytes per pass
            roofline_nn = {"bound": "hbm", "kernel": "k_icp_tile + k_icp_ball + k_icp_walk + k_icp_sums + k_reduce_sums (one ICP pass, this rank's share)",
                           "algorithmic_bytes_per_pass": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "first_pass": {"ms": pass_ms[0], "achieved": alg / (pass_ms[0] * 1e-3) / 1e9,
                                          "frac": alg / (pass_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "converged": {"ms": conv_ms, "achieved": alg / (conv_ms * 1e-3) / 1e9,
                                         "frac": alg / (conv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "queries_per_s": (hi - lo) / (conv_ms * 1e-3)},
                           "traffic": measured_nn_traffic(), "valu_issue_frac_offline": measured_nn_valu_issue()}
            ctx2.close()
            del tgt_s, src_all
        except Exception as exc:      # an extra leg must not take the headline line with it
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["icp_1m"] = repr(exc)[:300]

        try:
            # configs[4]: the large scan-to-model pair (same generator, seeds 44 / 45), queries sharded over the ranks,
            # target replicated; ten passes from the 10-degree start
            if args.big_points:
                nb = args.big_points
                tgt_b = synth_surface(nb, 44)
                lo_b, hi_b = rank * nb // world, (rank + 1) * nb // world
                src_b = synth_surface(nb, 45, noise=1e-3)[lo_b:hi_b].astype(np.float64)
                src_b = (src_b @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
                ctx3 = Context(local)
                t0 = time.perf_counter()
                ctx3.set_target(torch.from_numpy(tgt_b).to(dev))
                torch.cuda.synchronize()
                extra["icp_16m_grid_build_s"] = time.perf_counter() - t0
                ctx3.set_source(torch.from_numpy(np.ascontiguousarray(src_b)).to(dev))
                ctx3.set_stream(torch.cuda.current_stream().cuda_stream)
                del tgt_b, src_b
                Tb = np.eye(4); pass_ms = []
                barrier(); t0 = time.perf_counter()
                for _ in range(10):
                    Tb = icp_iter(Tb, nb, ctx3)
                barrier()
                el = max_over_ranks(time.perf_counter() - t0)
                extra["icp_16m_points"] = nb
                extra["icp_16m_iterations_per_s"] = 10 / el
                extra["icp_16m_queries_per_s"] = nb * 10 / el
                extra["icp_16m_pass_ms"] = [round(x, 3) for x in pass_ms]
                extra["icp_16m_rot_err_deg"] = float(np.rad2deg(np.arccos(np.clip((np.trace(Tb[:3, :3] @ Tm[:3, :3]) - 1) / 2, -1, 1))))
                ctx3.close()
        except Exception as exc:      # an extra leg must not take the headline line with it
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["icp_16m"] = repr(exc)[:300]

        try:
            # configs[0], the reference's own run: bunny000 -> bunny045, 30 iterations through the host class
            # (src/main.cpp:270-284); every rank runs the whole registration
            from icp_amd.host import SingleThreadIcp
            st = SingleThreadIcp(load_scan("bunny000"), load_scan("bunny045"), iters=30, device=local)
            st.registration()
            barrier(); t0 = time.perf_counter()
            for _ in range(5):
                st.registration()
            barrier()
            extra["icp_bunny045_30it_ms"] = max_over_ranks(time.perf_counter() - t0) / 5 * 1e3
            del st
        except Exception as exc:      # an extra leg must not take the headline line with it
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["icp_bunny"] = repr(exc)[:300]

        try:
            # configs[3]: Go-ICP on the reference's bunny pair (model bunny000 = bun315/bun000, data bun315/bun270), both
            # clouds normalised isotropically into [-1,1]^3, DT 300^3, mse 1e-3. The 2 x 8 inner searches of a rotation cube
            # advance in lock-step, one launch per step (each starts from the optError of the moment the cube is popped,
            # where the reference's one-after-the-other order lets a later search see an earlier one's improvement), on the
            # device from start to end: queues and decisions stay there, the host enqueues steps (--goicp-host-search: a host
            # round trip per step, as before). `speculate` (and, host-driven, `lookahead`) only put more cubes into a
            # launch, the decisions are those of lookahead = speculate = 0. With several ranks the rotation cubes of a step (the popped one and those searched ahead with it) are
            # dealt out: each rank runs the inner searches of its cubes on its own GPU without communication, and one
            # all-reduce per step (RCCL on the context's stream, a few KB) spreads the results; every rank takes the
            # same decisions. --goicp-share-launches restores the earlier scheme (every launch cut into shares).
            if not args.no_goicp:
                from icp_amd.host import GoIcp, normalize_goicp
                gm, gd, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
                kw = {}
                if world > 1 and args.backend == "nccl":
                    kw = dict(allreduce_dev=lambda t: dist.all_reduce(t), dev_buffer=torch.zeros(1 << 18, dtype=torch.float64, device=dev))
                elif world > 1:
                    def allreduce_host(a):
                        t = torch.from_numpy(a); dist.all_reduce(t)
                    kw = dict(allreduce=allreduce_host)
                go = GoIcp(gm, gd, mse=1e-3, dt_div=DIV, device=local, rank=rank, world=world, **kw)
                go.options(batched=True, frontier_width=1, lookahead=args.goicp_lookahead, speculate=args.goicp_speculate,
                           shard_cubes=not args.goicp_share_launches, device_search=not args.goicp_host_search)
                barrier(); t0 = time.perf_counter()
                gr = go.registration()
                barrier()
                el = max_over_ranks(time.perf_counter() - t0)
                extra["goicp_bunny270"] = {
                    "time_to_optimum_s": el, "optimal_sse": float(gr["optError"]), "lower_bound": float(gr["lb"]),
                    "epsilon": float(gr["sseThresh"]), "initial_sse": float(gr["initError"]), "icp_sse": float(gr["icpError"]),
                    "rotation_cubes": gr["rot_pops"], "bound_evaluations": gr["inner_evals"], "launches": gr["launches"],
                    "points": int(gd.shape[0]), "order": "inner searches of a rotation cube in lock-step (batched)",
                    "inner_searches": "on the device (icpx_bnb_search_run)" if gr["search_steps"] else "driven by the host (icpx_bnb_submit per step)",
                    "search_steps": gr["search_steps"],
                    "lookahead": args.goicp_lookahead if not gr["search_steps"] else None, "speculate": args.goicp_speculate,
                    "searched_ahead_used": gr["spec_hits"], "searched_ahead_discarded": gr["spec_misses"],
                    "collectives": gr["collectives"],
                    "collective": "none" if world == 1 else
                                  (("device all-reduce (RCCL)" if args.backend == "nccl" else "host all-reduce") +
                                   (" per launch (every batch cut into shares)" if args.goicp_share_launches else
                                    " per step (rotation cubes dealt out to the ranks)"))}
                del go
        except Exception as exc:      # an extra leg must not take the headline line with it
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["goicp"] = repr(exc)[:300]

        try:
            # the bound kernel by itself: 512 families of eight sibling translation cubes (4 096 cubes) under random
            # rotations, lower-bound form at rotation level 3, against the full data cloud; events on the stream it runs on.
            # Algorithmic bytes: 16 B per data point once per family + the 4-byte gather per (cube, point) (the distance as
            # a float out of the brick copy of the grid). What bounds the kernel is neither HBM nor the vector units but the
            # rate at which the L1 takes lane addresses (one a clock per CU; the 64 points of a wave fall into some 39
            # different voxels per child cube): l1_accesses_per_clk_cu_offline, from the counters of a full Go-ICP run.
            if not args.no_goicp:
                import ctypes as C
                from icp_amd.host import normalize_goicp
                gm, gd, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
                ctx4 = Context(local)
                ctx4.set_target(gm); ctx4.edt_build(DIV); ctx4.goicp_set_data(gd)
                rng = np.random.default_rng(5)
                fam = 512
                Rs, cubes, levels = [], [], []
                for _ in range(fam):
                    v = rng.uniform(-1.5, 1.5, 3); th = float(np.linalg.norm(v)); a = v / th
                    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                    R = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32).reshape(9)
                    x, y, z = rng.uniform(-0.5, 0.25, 3); w = 0.25
                    for j in range(8):
                        Rs.append(R); levels.append(3)
                        cubes.append([x + (j & 1) * w / 2, y + (j >> 1 & 1) * w / 2, z + (j >> 2 & 1) * w / 2, w / 2])
                items = ctx4.make_items(Rs, np.asarray(cubes, np.float32), levels)
                nit = len(items)
                ctx4.bnb_eval(items)                                   # also makes the brick copies of the grid
                raw = np.frombuffer(C.string_at(C.addressof(items), C.sizeof(items)), np.uint8).copy()
                items_dev = torch.from_numpy(raw).to(dev)
                out_dev = torch.zeros(2 * nit, dtype=torch.float64, device=dev)
                ctx4.set_stream(torch.cuda.current_stream().cuda_stream)
                ctx4.bnb_eval_noalloc(items_dev, nit, out_dev); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                times = []
                for _ in range(10):
                    e0.record(); ctx4.bnb_eval_noalloc(items_dev, nit, out_dev); e1.record(); torch.cuda.synchronize()
                    times.append(e0.elapsed_time(e1))
                bn_ms = float(np.median(times))
                ndp = int(gd.shape[0])
                alg_b = 4.0 * nit * ndp + 16.0 * ndp * fam
                pm = _newest_profile("r*_goicp_pmc.json") or {}
                k8 = next((v for k, v in pm.items() if (k.startswith("k_search_eval_step") or k.startswith("k_bnb_eval8")) and isinstance(v, dict)), None)
                cyc = k8["GRBM_GUI_ACTIVE"] / 8.0 if k8 and k8.get("GRBM_GUI_ACTIVE") else None
                roofline_bnb = {"bound": "L1 lane-address rate (not hbm): see l1_accesses_per_clk_cu_offline", "kernel": "k_bnb_eval8 + k_bnb_reduce", "cubes": nit, "points": ndp,
                                "algorithmic_bytes_per_launch": alg_b, "avg_launch_ms": bn_ms, "achieved": alg_b / (bn_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_b / (bn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "lookups_per_s": nit * ndp / (bn_ms * 1e-3), "ns_per_cube": bn_ms * 1e6 / nit, "traffic": None,
                                # vector instructions issued x 4 cycles over the cycles of all 1 024 SIMDs, from the counters of a full
                                # Go-ICP run under rocprofv3 (offline evidence, profiles/)
                                "valu_issue_frac_offline": (k8["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc if cyc and "SQ_INSTS_VALU" in k8 else None),
                                # accesses the L1s of the 256 CUs took per clock and CU over the same run (1 is what an L1 can take)
                                "l1_accesses_per_clk_cu_offline": (k8["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc
                                                                   if cyc and "TCP_TOTAL_CACHE_ACCESSES_sum" in k8 else None)}
                ctx4.close()
        except Exception as exc:
            import traceback; traceback.print_exc()
            extra.setdefault("errors", {})["roofline_bnb"] = repr(exc)[:300]

    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as O
        mn, cell = O.ldt_box(tgt, DIV)
        occ, _ = O.ldt_splat(tgt, DIV, mn, cell)
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))

        def edt_rate(th, budget_s, max_reps):
            O.edt_meijster(occ, threads=th)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s and reps < max_reps:
                O.edt_meijster(occ, threads=th); reps += 1
            return voxels * reps / (time.perf_counter() - t0), reps

        v, reps = edt_rate(threads, 8.0, 40)
        v2, reps2 = edt_rate(2, 6.0, 10)
        cpu = {"value": v, "unit": "voxels/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
               "sample": f"{reps} builds of the same 300^3 bunny045 grid (three-pass EDT restatement, OpenMP)",
               "threads2": {"value": v2, "unit": "voxels/s", "cores": 2,
                            "sample": f"{reps2} builds, 2 threads (the reference's hard-coded thread count)"}}
        if not args.no_extra:   # the CPU restatement of configs[0]: kd-tree search like the reference's, queries spread over the threads
            t0 = time.perf_counter()
            O.icp(load_scan("bunny000"), tgt, 30, nn_mode=0, threads=threads)
            cpu["icp_bunny045_30it_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            O.icp(load_scan("bunny000"), tgt, 30, nn_mode=0, threads=2)
            cpu["threads2"]["icp_bunny045_30it_ms"] = (time.perf_counter() - t0) * 1e3

    watchdog.cancel()
    emit()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
