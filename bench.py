#!/usr/bin/env python3
"""bench.py -- stereo frame-pairs/s of the matcher hot path (pushBack + matchFeatures(2)) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one 200-frame synthetic KITTI-shaped stereo sequence
(BASELINE.json configs[1]: 1242x375, default matcher parameters) with the Tr_delta feedback of the
reference's stereo VO replayed from tests/golden/cfg4_seq200_tr_8seeds.npz.  Images are resident in
HBM before the timed region starts.  Two ways through the C-ABI are timed:
  * "value": the look-ahead entry point vsm_sequence_run (same results as frame-by-frame calls;
    the frames of a chunk share one launch per kernel, host stages run frame-parallel);
  * "per_frame_api": vsm_push_back_device + vsm_match once per frame, i.e. what the drop-in
    Matcher::pushBack / matchFeatures does inside a live VO loop.  With N > 1 every rank owns one independent sequence (seed 1234+rank) on its
own GPU; the only collectives are the barrier and the max-reduction of the elapsed time
(RCCL, a few bytes): weak scaling, no data-path exchange.

One JSON line is printed by rank 0 (see the contract in the task description) with two extra
objects: "roofline" (dominant kernel: algorithmic bytes per launch / HIP-event duration on the
library's own stream, vs 8 TB/s) and "cpu_baseline" (the reference -- oracle/_ref, built from the
reference's own sources -- or, if that build is absent, the scalar oracle port, on one host core).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything initialises HIP: see visomatch.py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "opencl-structure-from-motion_amd"
W, H, N_FRAMES = 1242, 375, 200
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def algorithmic_bytes(kernel, counts):
    """SURVEY.md section 8(d) per-unit figures, split per kernel.  counts: per-launch work."""
    P, Ph, Hh = counts["P"], counts["Ph"], counts["Hh"]
    if kernel == "k_filters<true>":      # read image, write du_full + dv_full
        return counts["imgs"] * (W * H + 2 * P * H)
    if kernel == "k_filters<false>":     # read half image, write du,dv (u8) + f1,f2 (i16)
        return counts["imgs"] * (Ph * Hh + 2 * Ph * Hh + 4 * Ph * Hh)
    if kernel.startswith("k_match"):      # 48 B/query record, 8 B/candidate, 32 B/SAD, 48 B/raw result
        return 48 * counts["Q"] + 8 * counts["C"] + 32 * counts["S"] + 48 * counts["Mraw"]
    if kernel == "k_refine":             # 3 relocations x 26 descriptors x 16 B per match
        return 1248 * counts["M"]
    if kernel.startswith("k_nms"):       # f1,f2 read once per set
        return counts["imgs"] * (4 * Ph * Hh)
    if kernel == "k_emit":               # 32 B descriptor gather + 48 B record per feature
        return counts["imgs"] * 80 * counts.get("N", 0)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=N_FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-per-frame", action="store_true", help="skip the per-frame API leg (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    shard = importlib.import_module(PKG + ".shard")
    rank, local_rank, world = shard.rank_info()
    if world != args.gpus:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using {world}", file=sys.stderr)
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev        # (more ranks than GPUs only happens in rehearsals)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm_dev = dev
    if world > 1:
        if world <= ndev:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI: one rank per GPU
        else:
            # rehearsal on a box with fewer GPUs than ranks (RCCL refuses two ranks on one GPU):
            # the few bytes of aggregation go over gloo instead
            dist.init_process_group("gloo")
            comm_dev = torch.device("cpu")

    # host threads for the exact Delaunay stage: this rank's share of the CPUs the job may use
    # (cgroup quota if there is one -- more spinning threads than that only get throttled)
    ncpu = os.cpu_count() or 16
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            ncpu = min(ncpu, max(1, int(q) // int(per)))
    except Exception:
        pass
    os.environ.setdefault("VSM_HOST_THREADS", str(max(2, min(32, ncpu // max(world, 1)))))
    vm = importlib.import_module(PKG + ".visomatch")
    synth = importlib.import_module(PKG + ".synth")
    vm.lib()  # raises if the HIP library is missing

    # ---- this rank's sequence, resident in HBM -------------------------------------------
    seed = shard.sequence_seed(rank)
    nf = args.frames
    cv = synth.canvas(seed, W, H)
    host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])  # [F,2,H,W]
    frames = torch.from_numpy(host).to(dev)
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
    key = f"s{seed}" if f"s{seed}_tr_in" in g.files else "s1234"
    tr_in, tr_valid = g[key + "_tr_in"], g[key + "_tr_valid"]
    intr = [float(x) for x in g["intr"]]

    m = vm.Matcher()
    m.set_intrinsics(*intr)

    left_d, right_d = frames[:, 0], frames[:, 1]          # [F,H,W] views, resident in HBM
    tr12 = np.ascontiguousarray(tr_in[:nf].reshape(nf, 16)[:, :12])
    trv = np.ascontiguousarray(tr_valid[:nf].astype(np.uint8))

    def run_frames(collect=None):                           # drop-in per-frame path
        for f in range(nf):
            rc = m.push_back(frames[f, 0], frames[f, 1])
            assert rc == 0, rc
            m.match_features(2, tr_in[f] if tr_valid[f] else None)
            if collect is not None:
                collect.append(m.get_matches())

    def run_sequence(collect=None):                         # look-ahead path
        # like the per-frame loop above, the timed call leaves the lists inside the matcher
        # (getMatches() copies are taken only by the verification pass)
        m.run_sequence(left_d, right_d, 2, tr12, trv, fetch=False)
        if collect is not None:
            collect.extend(m.sequence_matches(f) for f in range(nf))

    dmod = dist if world > 1 else None

    for _ in range(args.warmup):
        run_sequence()
    shard.barrier(dmod, comm_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        run_sequence()
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    torch.cuda.synchronize()
    shard.barrier(dmod, comm_dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    total_pairs, elapsed, _ = shard.aggregate(dmod, torch, nf * args.steps, elapsed, comm_dev)
    value = total_pairs / elapsed
    seq_t = m.sequence_timings()          # phase split of the last timed step (rank-local)

    # ---- the same sequence through the per-frame (drop-in) entry points ---------------------
    per_frame_value = None
    if not args.no_per_frame:
        run_frames()
        shard.barrier(dmod, comm_dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_frames()
        torch.cuda.synchronize()
        pf_elapsed = time.perf_counter() - t1
        pf_pairs, pf_elapsed, _ = shard.aggregate(dmod, torch, nf, pf_elapsed, comm_dev)
        per_frame_value = pf_pairs / pf_elapsed

    # ---- the whole VisualOdometryStereo::process loop with LIVE Tr_delta feedback (row f-2) ---
    vo_value, vo_ok = None, None
    if not args.no_per_frame:
        def run_vo(check=None):
            vm.vo_sampler_seed(71)                      # what a fresh process of the reference starts from
            vo = vm.VisualOdometryStereo(*intr)
            torch.cuda.synchronize()
            t = time.perf_counter()
            trail = [vo.process(frames[f, 0], frames[f, 1])[3] for f in range(nf)]
            dt = time.perf_counter() - t
            vo.close()
            return dt, np.array(trail)
        run_vo()
        shard.barrier(dmod, comm_dev)
        vo_dt, trail = run_vo()
        vo_pairs, vo_dt, _ = shard.aggregate(dmod, torch, nf, vo_dt, comm_dev)
        vo_value = vo_pairs / vo_dt
        if seed == 1234 and not args.no_verify:
            ge = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_seq200_ego.npz"))
            vo_ok = bool(trail.tobytes() == ge["tr_out"][:nf].tobytes())

    # ---- monocular egomotion (row f-4): 2000 hypotheses on a synthetic 5000-match scene -----------
    mono = None
    if rank == 0 and not args.no_per_frame:
        try:
            mono = mono_leg(vm, not args.no_cpu_baseline)
        except Exception as e:  # informational leg: never takes the headline down with it
            mono = {"error": repr(e)}

    # ---- verification (outside the timed region): final lists vs the reference's hashes -----
    verified = None
    if not args.no_verify and key == f"s{seed}":
        lists, lists_pf = [], []
        run_sequence(lists)
        m.close()
        m = vm.Matcher()           # fresh ring buffer: frame 0 has no predecessor, like the fixture
        m.set_intrinsics(*intr)
        if not args.no_per_frame:
            run_frames(lists_pf)
        else:
            lists_pf = lists
        ok = all(len(l[f]) == int(g[key + "_counts"][f]) and sha(l[f]) == str(g[key + "_hashes"][f])
                 for f in range(nf) for l in (lists, lists_pf))
        _, _, verified = shard.aggregate(dmod, torch, 0, 0, comm_dev, all_ok=bool(ok))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline leg: per-kernel HIP-event durations over one more sequence --------------
    m.set_profiling(True)
    run_sequence()
    torch.cuda.synchronize()
    stats = m.kernel_stats()
    m.set_profiling(False)
    # dominant kernel on the HBM side (k_export_list is a PCIe copy into pinned host memory)
    dom = max((k for k in stats if k != "k_export_list"), key=lambda k: stats[k][0])
    dom_ms, dom_n = stats[dom]
    # per-launch work counters of the dominant kernel from one representative frame pair
    # (features/candidates are stationary over this sequence)
    P, Ph, Hh = W + 15 - (W - 1) % 16, (W // 2) + 15 - ((W // 2) - 1) % 16, H // 2
    chunk = int(m.sequence_timings()["chunk"]) or 1
    # (a kernel runs once per chunk; chunks are not all the same size - a long sequence starts and ends with a half
    # chunk - so the per-launch figures below are averages over the launches HIP events were taken of: the whole
    # sequence's work divided by the kernel's own launch count)
    counts = dict(P=P, Ph=Ph, Hh=Hh)
    cpu = None
    work = None
    if not args.no_cpu_baseline:
        cpu, work = cpu_baseline(host, tr_in, tr_valid, intr)
    # HBM traffic per launch from the committed rocprofv3 PMC summary of this same command
    # (tools/pmc_summary.py; PMC passes cannot run inside the timed process)
    pmc = {}
    try:
        import csv
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_lookahead_pmc_hbm.csv"))):
            pmc[r["bench_name"]] = int(float(r["traffic_bytes_per_launch"]))
    except Exception:
        pass

    def roofline_of(kname):
        ms, nl = stats[kname]
        if nl == 0:
            return None
        c2 = dict(counts, imgs=2.0 * nf / nl)
        if work is not None:
            pairs = (nf - 1) / nl
            w1 = kname.endswith("pass1")
            c2.update(Q=work["Q1" if w1 else "Q2"] * pairs, C=work["C1" if w1 else "C2"] * pairs,
                      S=work["S1" if w1 else "S2"] * pairs, Mraw=work["M1" if w1 else "M"] * pairs, M=work["M"] * pairs,
                      N=work["N"])
        elif kname.startswith(("k_match", "k_refine", "k_emit")):
            return None
        ab = algorithmic_bytes(kname, c2)
        if ab is None or nl == 0:
            return None
        achieved = ab / (ms / nl / 1e3) / 1e9
        return dict(bound="hbm", kernel=kname, achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6), traffic=pmc.get(kname), avg_launch_us=round(ms / nl * 1e3, 3),
                    algorithmic_bytes_per_launch=int(ab))

    roof = roofline_of(dom)
    roof_all = {k: r for k in stats if stats[k][1] for r in [roofline_of(k)] if r}
    out = {
        "metric": "stereo frame-pairs/sec (1242x375) through pushBack+matchFeatures(2), p_matched bit-exact",
        "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"KITTI-shaped synthetic stereo sequence 1242x375, {nf} frames per GPU, quad matching, "
                               "default parameters, replayed Tr_delta feedback, look-ahead C-ABI entry point "
                               f"vsm_sequence_run (chunks of {chunk} frames)",
                   "frames_per_step": nf, "sequences": world, "inputs": "resident in HBM"},
        "per_frame_api": {"value": round(per_frame_value, 3) if per_frame_value else None, "unit": "frame-pairs/s",
                          "what": "same sequence through vsm_push_back_device + vsm_match per frame (drop-in "
                                  "Matcher::pushBack/matchFeatures path)"},
        "vo_process_api": {"value": round(vo_value, 3) if vo_value else None, "unit": "frames/s",
                           "tr_delta_trail_bit_exact_vs_reference": vo_ok,
                           "what": "vsm_vo_stereo_process_device per frame: pushBack + matchFeatures(2, live Tr_delta) + "
                                   "bucketFeatures + RANSAC/Gauss-Newton egomotion (VisualOdometryStereo::process)"},
        "vo_mono_egomotion": mono,
        "verified_bit_exact_vs_reference_hashes": verified,
        "roofline": roof,
        "roofline_by_kernel": {k: {"achieved_GBps": r["achieved"], "frac": r["frac"], "avg_launch_us": r["avg_launch_us"]}
                               for k, r in roof_all.items()},
        "cpu_baseline": cpu,
        "kernel_ms_per_frame": {k: round(v[0] / nf, 5) for k, v in stats.items() if v[1]},
        "kernel_avg_launch_us": {k: round(v[0] / v[1] * 1e3, 2) for k, v in stats.items() if v[1]},
        "sequence_timings_us": seq_t,
        # what the main stream's kernels alone would sustain: frames / their summed HIP-event durations in the profiled pass
        "gpu_phases_only_frame_pairs_per_s": round(nf / (sum(v[0] for v in stats.values()) * 1e-3), 1),
        "host_threads": int(os.environ["VSM_HOST_THREADS"]),
        "step_ms_rank0": step_ms,
        "match_timings_us_last_frame": m.timings(),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def mono_leg(vm, with_cpu):
    """VisualOdometryMono::estimateMotion (the path the reference offloads to OpenCL) on flow matches of
    a synthetic ground-plane scene: HIP kernels for the hypothesis fits, Sampson inlier counts,
    triangulation and plane vote.  Checked against the oracle restatement of the reference's CPU class."""
    rs = np.random.RandomState(7)
    n, iters = 5000, 2000
    f, cu, cv = 721.5377, 609.5593, 172.854
    ng = n // 2
    X = np.concatenate([rs.uniform(-8, 8, ng), rs.uniform(-10, 10, n - ng)])
    Y = np.concatenate([np.full(ng, 1.65), rs.uniform(-3, 1.2, n - ng)])
    Z = np.concatenate([rs.uniform(4, 40, ng), rs.uniform(5, 50, n - ng)])
    ry, tz = 0.012, -0.9
    P = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]]) @ np.stack([X, Y, Z])
    P[2] += tz
    m = np.zeros(n, dtype=vm.P_MATCH)
    m["u1p"], m["v1p"] = f * X / Z + cu, f * Y / Z + cv
    m["u1c"], m["v1c"] = f * P[0] / P[2] + cu, f * P[1] / P[2] + cv
    for k in ("u1p", "v1p", "u1c", "v1c"):
        m[k] += rs.normal(0, 0.1, n).astype(np.float32)
    bad = rs.permutation(n)[: n // 5]
    for k in ("u1c", "v1c"):
        m[k][bad] += rs.uniform(-30, 30, len(bad)).astype(np.float32)
    for k in ("u2p", "v2p", "u2c", "v2c"):
        m[k] = -1
    kw = dict(height=1.65, pitch=-0.08, ransac_iters=iters)
    v = vm.VisualOdometryMono(f, cu, cv, **kw)
    best = 1e9
    for _ in range(5):
        vm.vo_sampler_seed(71)
        t = time.perf_counter()
        ok, T = v.process_matches(m)
        best = min(best, time.perf_counter() - t)
    inl = v.get_inlier_indices()
    out = {"matches": n, "ransac_iters": iters, "ms_per_estimate": round(best * 1e3, 3), "success": bool(ok),
           "inliers": int(len(inl)), "fits_and_triangulation_on_gpu": v.device_svd(),
           "split_us": {k: round(float(x), 1) for k, x in zip(("fits", "inlier_count", "rt_triangulation", "plane_vote"),
                                                              v.timings()[4:8])}}
    v.close()
    if with_cpu:
        from oracle import bindings as B
        B.oracle_sampler_seed(71)
        o = B.OracleMonoVO(f, cu, cv, **kw)
        t = time.perf_counter()
        ok_o, T_o = o.process_matches(m)
        dt = time.perf_counter() - t
        out["cpu_ms_per_estimate"] = round(dt * 1e3, 2)
        out["cpu_kind"] = "port (oracle restatement of the reference's CPU class, 1 thread)"
        out["bit_exact_vs_cpu"] = bool(ok_o == ok and T_o.tobytes() == T.tobytes() and np.array_equal(o.inliers(), inl))
        o.close()
    return out


def cpu_baseline(host, tr_in, tr_valid, intr, budget_s=20.0):
    """CPU leg on rank 0: the real reference (oracle/_ref) if its build travelled here, else the
    scalar oracle port; one thread, a bounded prefix of the same sequence.  Also returns the
    per-frame-pair work counters (oracle only) that price the match kernels' algorithmic bytes."""
    from oracle import bindings as B
    kind = "reference" if B.have_ref() else "port"
    cm = B.CpuMatcher("ref" if kind == "reference" else "oracle")
    cm.set_intrinsics(*intr)
    t0 = time.perf_counter()
    n = 0
    for f in range(host.shape[0]):
        cm.push_back(host[f, 0], host[f, 1])
        cm.match(2, tr_in[f] if tr_valid[f] else None)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cm.close()
    # work counters of one representative frame pair (frames 1,2) from the oracle
    om = B.CpuMatcher("oracle")
    om.set_intrinsics(*intr)
    work = None
    for f in range(3):
        om.push_back(host[f, 0], host[f, 1])
        om.match(2, tr_in[f] if tr_valid[f] else None)
    c = om.counters()
    work = dict(Q1=c["Q1"], C1=c["C1"], S1=c["S1"], Q2=c["Q"] - c["Q1"], C2=c["C"] - c["C1"], S2=c["S"] - c["S1"],
                M1=len(om.stage(0)), M=int(c["M"]), N=len(om.features("1c1")) + len(om.features("1c2")))
    om.close()
    vo_fps = None
    if kind == "reference":
        rv = B.RefStereoVO(*intr)
        t1 = time.perf_counter()
        k = 0
        for f in range(host.shape[0]):
            rv.process(host[f, 0], host[f, 1])
            k += 1
            if time.perf_counter() - t1 > budget_s / 2:
                break
        vo_fps = round(k / (time.perf_counter() - t1), 3)
        rv.close()
    return (dict(value=round(n / dt, 3), unit="frame-pairs/s", cores=1, kind=kind,
                 sample=f"first {n} frames of the same sequence, pushBack+matchFeatures(2), 1 thread",
                 vo_process_frames_per_s=vo_fps), work)


if __name__ == "__main__":
    main()
