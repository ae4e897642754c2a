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
import gc
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np



def _cpu_share():
    """this rank's share of the CPUs the job may use (cgroup quota if there is one - more spinning threads than that only get
    throttled) and the host-pool size that follows from it"""
    ncpu = os.cpu_count() or 16
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            ncpu = min(ncpu, max(1, int(q) // int(per)))
    except Exception:
        pass
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    share = ncpu // max(world, 1)
    # (two CPUs of the share are left to the HIP runtime's own threads: with as many pool threads as CPUs the rank runs into
    # its quota now and then, and a throttled call takes 10-14 ms instead of 5)
    return ncpu, max(2, min(32, share - 2 if share >= 8 else share))


def _near_gpu():
    """This rank's threads - Python's, the HIP runtime's, the library's host pool - onto the CPUs of its GPU's NUMA node, BEFORE
    anything initialises HIP: the runtime allocates its kernel-argument and signal pools in host memory when it starts, on the
    node of the thread that starts it, and an MI355X node has two sockets with four GPUs each.  A process on the far socket runs
    the look-ahead call in 4.45-4.57 ms, on the near one in 4.17-4.23, left to the scheduler in anything between - process by
    process (tools/numa_probe.sh).  What `numactl --cpunodebind` per rank does in a launch script, done here so that the
    driver's plain `python bench.py` / torchrun lines get it too.  The GPU of local rank r = the r-th render node this
    container sees; its CPUs from sysfs, no HIP call."""
    try:
        import glob
        import re
        nodes = sorted((int(re.sub(r"\D", "", os.path.basename(n))), os.path.basename(n)) for n in glob.glob("/dev/dri/renderD*"))
        nodes = [n for _, n in nodes if open(f"/sys/class/drm/{n}/device/vendor").read().strip() == "0x1002"]
        if not nodes:
            return None
        name = nodes[int(os.environ.get("LOCAL_RANK", "0")) % len(nodes)]
        cpus = set()
        for tok in open(f"/sys/class/drm/{name}/device/local_cpulist").read().strip().split(","):
            a, _, b = tok.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        both = cpus & allowed
        if len(both) >= 2 and len(both) < len(allowed):
            os.sched_setaffinity(0, both)
            return {"render_node": name, "numa_node": int(open(f"/sys/class/drm/{name}/device/numa_node").read()), "cpus": len(both)}
    except Exception:
        pass
    return None


NEAR_GPU = _near_gpu()
NCPU, _threads = _cpu_share()
os.environ.setdefault("VSM_HOST_THREADS", str(_threads))
# Before anything initialises HIP (the runtime reads it once).  FIVE hardware queues - the null stream's, the handle's main
# stream's and three side streams' - carry the look-ahead path, and a sixth queue in the process throttles every kernel's
# workgroup dispatch (DESIGN_HISTORY.md section 6c), so five is also the cap: whatever else creates streams here (RCCL for the
# start / end reductions) shares a queue instead of adding one.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")

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
    W, H = counts.get("W", globals()["W"]), counts.get("H", globals()["H"])
    if kernel == "k_filters<true>":      # read image, write du_full + dv_full
        return counts["imgs"] * (W * H + 2 * P * H)
    if kernel == "k_front":              # read the caller's image once; write half image + the tiled du/dv plane (2 B / pixel)
        return counts["imgs"] * (W * H + Ph * Hh + 2 * P * H)
    if kernel == "k_filters<false>":     # read half image, write du,dv (u8) + f1,f2 (i16)
        return counts["imgs"] * (Ph * Hh + 2 * Ph * Hh + 4 * Ph * Hh)
    if kernel == "k_feat_dense":         # filters + dense suppression out of one LDS tile: the work of k_filters<false> + k_nms:dense
        return counts["imgs"] * (Ph * Hh + 2 * Ph * Hh + 4 * Ph * Hh + 4 * Ph * Hh)   # (f1 / f2 written once + read once per SURVEY 8(d); they stay in LDS)
    if kernel == "k_feat_sparse":        # the sparse scale's suppression: f1, f2 read once (recomputed from the half image instead)
        return counts["imgs"] * (4 * Ph * Hh)
    if kernel.startswith("k_match"):      # 48 B/query record, 8 B/candidate, 32 B/SAD, 48 B/raw result
        return 48 * counts["Q"] + 8 * counts["C"] + 32 * counts["S"] + 48 * counts["Mraw"]
    if kernel == "k_refine":             # 3 relocations x 26 descriptors x 16 B per match
        return 1248 * counts["M"]
    if kernel.startswith("k_nms"):       # f1,f2 read once per set
        return counts["imgs"] * (4 * Ph * Hh)
    if kernel == "k_emit":               # 32 B descriptor gather + 48 B record per feature
        return counts["imgs"] * 80 * counts.get("N", 0)
    if kernel == "k_dc_block" and "M" in counts:            # keys in, 32-byte triangle records + points + ids out
        return 80 * counts["M"]
    if kernel == "k_dc_merge" and "M" in counts:            # every level re-reads and re-writes the records near its seams: <= one pass
        return 2 * 64 * counts["M"]
    if kernel == "k_dc_prepare_kd_order" and "M" in counts: # keys in and out, 28 bytes of list scratch per point
        return 44 * counts["M"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=N_FRAMES)
    ap.add_argument("--startup", type=int, default=0, help="untimed calls in front of the warm-up steps (reported as startup_calls; none by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-per-frame", action="store_true", help="skip the per-frame API leg (profiling runs)")
    ap.add_argument("--no-alone", action="store_true", help="skip the non-overlapped profiling pass (runs under rocprofv3: its launches would mix into the averages)")
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

    # (the import-time guess of this rank's GPU - the r-th render node - checked against the device HIP really gave it: if the
    # guess was the other socket, at least the threads created from here on - the library's host pool - go to the right one)
    global NEAR_GPU
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        cpus = set()
        for tok in open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip().split(","):
            a, _, b = tok.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        if cpus and not (cpus & os.sched_getaffinity(0)):
            os.sched_setaffinity(0, cpus)
            NEAR_GPU = {"pci": bdf, "cpus": len(cpus), "note": "re-pinned after HIP start: the import-time guess was another socket"}
    except Exception:
        pass
    ncpu = NCPU   # (host threads for the exact Delaunay stage: sized at import time, before HIP initialises - see the top of the file)
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
        # like the per-frame loop above, the timed call leaves the lists inside the matcher - in the reference's 48-byte
        # p_match form, in host memory - (getMatches() copies are taken only by the verification pass)
        m.run_sequence(left_d, right_d, 2, tr12, trv, fetch=False)
        if collect is not None:
            collect.extend(m.sequence_matches(f) for f in range(nf))

    import contextlib

    @contextlib.contextmanager
    def beside_forkjoin():
        """The per-frame legs' calling thread inside the L3 domain of the library's fork-join threads (vsm_forkjoin_cpus,
        include/visomatch.h): it takes part in the one triangulation of every matchFeatures, and from another core complex every
        record it touches crosses between complexes.  Restored afterwards: the look-ahead legs want the whole node."""
        before = os.sched_getaffinity(0)
        dom = set(vm.forkjoin_cpus()) & before
        if dom and os.environ.get("VSM_BENCH_PIN_CALLER", "1") != "0":
            os.sched_setaffinity(0, dom)
        try:
            yield sorted(dom)
        finally:
            os.sched_setaffinity(0, before)

    dmod = dist if world > 1 else None

    # The interpreter's cyclic garbage collector stays off while the legs below are timed (as timeit does): with torch and numpy
    # loaded a full collection is a 55-65 ms pause of THIS harness, not of the library - it fell on frame 74 of the live VO leg's
    # second pass in every run (tools/vo_timing.py: 0.95 ms per frame with it, 0.62-0.71 without), and a collection of the young
    # generations is the 2 ms call the per-frame legs showed now and then.  Collections run between the legs instead.
    gc.collect()
    gc.freeze()
    gc.disable()

    # (rounds 3-4 let 30 calls pass here: once per process one call took 10-13 ms.  Its cause - the runtime creating a hardware
    # queue in the middle of a run for a table upload whose DMA engine was busy - is gone (k_upload), and so are
    # the hidden calls; --startup N still runs N untimed calls and the line reports them)
    startup_ms = []
    for _ in range(args.startup):
        ts = time.perf_counter()
        run_sequence()
        startup_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    for _ in range(args.warmup):
        ts = time.perf_counter()
        run_sequence()
        startup_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
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
    lookahead_form = m.sequence_path()
    # what every rank ran with (a SCALE run explains its own efficiency: DESIGN section 7): its host threads, the CPUs it is
    # confined to, frames per chunk, its own median and slowest step
    rank_rows = shard.gather_rows(dmod, torch, [rank, int(os.environ["VSM_HOST_THREADS"]), len(os.sched_getaffinity(0)), seq_t.get("chunk", 0),
                                                sorted(step_ms)[len(step_ms) // 2], max(step_ms), lookahead_form], comm_dev)
    ranks_info = [{"rank": int(r[0]), "host_threads": int(r[1]), "cpus_allowed": int(r[2]), "chunk_frames": int(r[3]), "median_step_ms": round(r[4], 3),
                   "slowest_step_ms": round(r[5], 3), "lookahead_form": int(r[6])} for r in rank_rows]

    gc.collect()
    # ---- the same sequence through the per-frame (drop-in) entry points ---------------------
    per_frame_value = None
    caller_cpus = None
    if not args.no_per_frame:
        with beside_forkjoin() as caller_cpus:
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
    VO_SKIP, vo_steady = 8, [None]
    if not args.no_per_frame:
        def run_vo(check=None):
            vm.vo_sampler_seed(71)                      # what a fresh process of the reference starts from
            vo = vm.VisualOdometryStereo(*intr)
            torch.cuda.synchronize()
            t = time.perf_counter()
            trail, stamps = [], []
            for f in range(nf):
                trail.append(vo.process(frames[f, 0], frames[f, 1])[3])
                stamps.append(time.perf_counter())
            dt = stamps[-1] - t
            vo.close()
            # (a new VO object is a new matcher handle: its first frames allocate the ring - 50 ms of the runtime's page-locked
            # allocations and thread starts that belong to the constructor, not to a frame; both rates are reported)
            vo_steady[0] = (nf - VO_SKIP) / (stamps[-1] - stamps[VO_SKIP - 1])
            return dt, np.array(trail)
        with beside_forkjoin():
            run_vo()
            gc.collect()
            shard.barrier(dmod, comm_dev)
            vo_dt, trail = run_vo()
        vo_pairs, vo_dt, _ = shard.aggregate(dmod, torch, nf, vo_dt, comm_dev)
        vo_value = vo_pairs / vo_dt
        if seed == 1234 and not args.no_verify:
            ge = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_seq200_ego.npz"))
            vo_ok = bool(trail.tobytes() == ge["tr_out"][:nf].tobytes())

    gc.collect()
    # ---- K sequences in lock-step with live feedback (row f-3, multi-sequence per GPU) -----------------
    multi = None
    if rank == 0 and world == 1 and not args.no_per_frame:
        try:
            multi = multi_sequence_leg(vm, synth, torch, dev, W, H)
        except Exception as e:  # informational leg: never takes the headline down with it
            multi = {"error": repr(e)}

    # ---- monocular egomotion (row f-4): 2000 hypotheses on a synthetic 5000-match scene -----------
    mono = None
    if rank == 0 and not args.no_per_frame:
        try:
            mono = mono_leg(vm, not args.no_cpu_baseline)
        except Exception as e:  # informational leg: never takes the headline down with it
            mono = {"error": repr(e)}

    # ---- verification (outside the timed region): final lists vs the reference's hashes -----
    # (every rank takes part in the reduction, whether it has a golden trail for its seed or not)
    verified, n_verified = None, 0
    if not args.no_verify:
        ok, have = True, key == f"s{seed}"
        if have:
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
        n_verified, _, verified = shard.aggregate(dmod, torch, 1 if have else 0, 0, comm_dev, all_ok=bool(ok))
        n_verified = int(n_verified)
        if n_verified == 0:
            verified = None

    # ---- the same look-ahead call fed from host memory (PCIe inclusive: the reference's pushBack takes host images) ----
    host_in_value, host_in_verified, host_in_calls = None, None, None
    host_pin_value, host_pin_verified = None, None
    if not args.no_per_frame:
        hl, hr = np.ascontiguousarray(host[:, 0]), np.ascontiguousarray(host[:, 1])
        m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
        if not args.no_verify:   # (outside the timed region, like the headline's check)
            if key == f"s{seed}":
                host_in_verified = bool(all(sha(m.sequence_matches(f)) == str(g[key + "_hashes"][f]) for f in range(nf)))
        shard.barrier(dmod, comm_dev)
        m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
        th = time.perf_counter()
        host_in_calls = []
        for _ in range(6):
            tc = time.perf_counter()
            m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
            host_in_calls.append(round((time.perf_counter() - tc) * 1e3, 2))
        hdt = time.perf_counter() - th
        hp, hdt, _ = shard.aggregate(dmod, torch, 6 * nf, hdt, comm_dev)
        host_in_value = hp / hdt
        # ... and once more from the SAME two arrays page-locked once by their owner (vsm_host_register + option seq_host_pinned:
        # what a caller that reuses its frame buffers can do; the copies leave straight from its memory)
        host_pin_value, host_pin_verified = None, None
        if vm.host_register(hl) and vm.host_register(hr):
            try:
                m.set_option("seq_host_pinned", 1)
                m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
                if not args.no_verify and key == f"s{seed}":
                    host_pin_verified = bool(all(sha(m.sequence_matches(f)) == str(g[key + "_hashes"][f]) for f in range(nf)))
                m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
                th = time.perf_counter()
                for _ in range(6):
                    m.run_sequence(hl, hr, 2, tr12, trv, fetch=False)
                host_pin_value = 6 * nf / (time.perf_counter() - th)
            finally:
                m.set_option("seq_host_pinned", 0)
                vm.host_unregister(hl)
                vm.host_unregister(hr)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- secondary configurations (BASELINE.json configs[2] and [4]): rank 0, N = 1 only, parity flag included ----
    secondary = None
    if world == 1 and not args.no_per_frame:
        try:
            secondary = secondary_configs(vm, synth, torch, dev)
        except Exception as e:  # informational legs never take the headline down with them
            secondary = {"error": repr(e)}

    # ---- roofline leg: per-kernel HIP-event durations over one more sequence --------------
    m.set_profiling(True)
    for _ in range(3):   # (how much of the Delaunay chains overlaps a kernel differs from call to call: three calls' launches)
        run_sequence()
    torch.cuda.synchronize()
    stats = {k: (ms / 3.0, n // 3) for k, (ms, n) in m.kernel_stats().items()}   # per call
    m.set_profiling(False)
    # ... and once more with nothing overlapping (VSM_SEQ_SERIAL=1: every group of launches drains before the next is
    # enqueued): the same kernels on the same data with the GPU to themselves.  In the timed region the matching kernels
    # share their SIMDs with the single-lane seam walks of the exact Delaunay chains, which take vector-issue cycles at the
    # price of a full wavefront; "alone" is what the kernel itself does, "frac" what it gets in the pipeline.
    stats_alone = None
    if not args.no_alone:
        m.set_option("seq_serial", 1)
        try:
            m.set_profiling(True)
            run_sequence()
            torch.cuda.synchronize()
            stats_alone = m.kernel_stats()
        finally:
            m.set_profiling(False)
            m.set_option("seq_serial", 0)
    # The hot path proper (filter + match + refinement kernels on the main stream) and, announced beside it, whatever leads
    # over ALL streams - the exact Delaunay kernels run on streams of their own (k_export_list is a PCIe copy, not HBM)
    hot = [k for k in stats if stats[k][1] and not k.startswith("k_dc_") and k != "k_export_list"]
    # (dominant = the longest with the GPU to itself when that pass was made: in the pipeline two kernels of similar length
    # swap places from run to run with how much of the chains happens to overlap them)
    dom_by = stats_alone if stats_alone else stats
    dom = max(hot, key=lambda k: dom_by.get(k, (0, 0))[0])
    dom_all = max((k for k in stats if stats[k][1] and k != "k_export_list"), key=lambda k: stats[k][0])
    gpu_total_ms = sum(v[0] for k, v in stats.items() if k != "k_export_list")
    dom_ms, dom_n = stats[dom]
    # The dominant kernel once more, in the pipeline, with spans around ITS launches only: a span is two event records on the
    # kernel's stream, and with every kernel of every stream bracketed the side streams are never empty of such packets -
    # which slows whatever runs beside them (tools/l2_invalidate_probe.py: bare event records on another stream, back to
    # back, take k_match from 177 to 219 us, k_refine from 100 to 155).  The roofline figure below is this measurement; the
    # all-kernels pass keeps its (perturbed) figure beside it.
    m.set_profiling(True, only=dom)
    for _ in range(3):
        run_sequence()
    torch.cuda.synchronize()
    dom_only = m.kernel_stats()[dom]
    m.set_profiling(False)
    dom_only = (dom_only[0] / 3.0, dom_only[1] // 3)
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
    # HBM traffic per launch: rocprofv3 PMC passes cannot run inside the timed process, so this figure comes from the
    # committed summary of the same command (tools/profile_all.sh -> tools/pmc_summary.py) and is labelled as such, with
    # the commit the summary was taken at; it is dropped if the library has been rebuilt since
    pmc, pmc_src = {}, None
    try:
        import csv
        import glob
        pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lookahead_pmc_hbm.csv")))[-1]  # (the latest round's)
        lib_path = os.path.join(ROOT, PKG, "libvisomatch.so")
        meta = {}
        for line in open(pmc_path):
            if line.startswith("#") and ":" in line:
                a, b = line[1:].split(":", 1)
                meta[a.strip()] = b.strip()
        rows = csv.DictReader(l for l in open(pmc_path) if not l.startswith("#"))
        for r in rows:
            pmc[r["bench_name"]] = int(float(r["traffic_bytes_per_launch"]))
        # (taken from the same kernel sources as the library that runs now?  file times say nothing after a checkout)
        csrc = os.path.join(ROOT, PKG, "csrc")
        hsh = hashlib.sha256()
        for pat in ("*.hip", "*.h", "*.inc", "*.cpp"):
            for fn in sorted(glob.glob(os.path.join(csrc, pat))):
                hsh.update(open(fn, "rb").read())
        pmc_src = {"file": "profiles/" + os.path.basename(pmc_path), "commit": meta.get("commit"), "taken": meta.get("taken"),
                   "same_kernel_sources": (meta.get("sources", "").split(" ")[0] == hsh.hexdigest()[:16]) if meta.get("sources") else None}
    except (FileNotFoundError, IndexError):
        pmc_src = {"file": None, "note": "no committed PMC summary"}
    except Exception as e:
        pmc_src = {"file": None, "note": repr(e)}

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
        r = dict(bound="hbm", kernel=kname, achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                 frac=round(achieved / HBM_PEAK_GBS, 6), traffic=pmc.get(kname), avg_launch_us=round(ms / nl * 1e3, 3),
                 algorithmic_bytes_per_launch=int(ab))
        if stats_alone and stats_alone.get(kname, (0, 0))[1]:
            ms1, nl1 = stats_alone[kname]
            a1 = ab * (nl / nl1) / (ms1 / nl1 / 1e3) / 1e9   # (same work per sequence; per-launch bytes follow the launch count)
            r["alone"] = {"avg_launch_us": round(ms1 / nl1 * 1e3, 3), "achieved": round(a1, 3), "frac": round(a1 / HBM_PEAK_GBS, 6),
                          "what": "same kernel, same data, nothing else on the GPU (VSM_SEQ_SERIAL=1 pass)"}
        return r

    roof_all = {k: r for k in stats if stats[k][1] for r in [roofline_of(k)] if r}
    roof = roofline_of(dom)
    if roof is not None and dom_only[1]:
        every = {"avg_launch_us": roof["avg_launch_us"], "achieved": roof["achieved"], "frac": roof["frac"],
                 "what": "the same kernel in the pass that brackets EVERY kernel of every stream with event records (the by-kernel table): "
                         "those records are packets on six streams and slow what they measure"}
        stats_dom_all = stats[dom]
        stats[dom] = dom_only
        roof = roofline_of(dom)
        stats[dom] = stats_dom_all
        roof["measured"] = ("HIP events around this kernel's launches only, on its stream, in the pipeline of the timed call (3 calls); since round 5 "
                            "the first chunk's block kernel (1760 workgroups that fill every compute unit) starts beside the second chunk's "
                            "launch of this kernel instead of behind another launch: the call is 5 % shorter and this launch ~70 us longer "
                            "(option seq_block_after_p2 = 1 separates the two again: profiles/r05_lookahead_bench_block_after_p2.json)")
        roof["with_every_kernel_profiled"] = every
    if roof is not None:
        roof["traffic_source"] = pmc_src
        da = roof_all.get(dom_all)
        roof["dominant_all_streams"] = {"kernel": dom_all, "share_of_gpu_kernel_time": round(stats[dom_all][0] / max(gpu_total_ms, 1e-9), 3),
                                        "avg_launch_us": round(stats[dom_all][0] / stats[dom_all][1] * 1e3, 2),
                                        "frac": da["frac"] if da else None,
                                        "what": "kernel with the largest summed duration over all streams of the profiled pass"}
        # B * fps / peak with B = the algorithmic bytes of one frame pair (oracle's work counters)
        if work is not None:
            Bp = 2 * (W * H + 2 * P * H + 10 * Ph * Hh + 80 * work["N"]) + 48 * (work["Q1"] + work["Q2"]) + \
                8 * (work["C1"] + work["C2"]) + 32 * (work["S1"] + work["S2"]) + 1248 * work["M"] + 48 * work["M"]
            gpu_only_fps = nf / (gpu_total_ms * 1e-3)
            roof["whole_path"] = {"bytes_per_frame_pair": int(Bp), "frac": round(Bp * value / world / (HBM_PEAK_GBS * 1e9), 6),
                                  "frac_gpu_kernels_only": round(Bp * gpu_only_fps / (HBM_PEAK_GBS * 1e9), 6),
                                  "what": "B * frame-pairs/s / 8 TB/s per GPU; gpu_kernels_only = with the summed kernel durations of all "
                                          "streams (PCIe export excluded) instead of the wall clock"}
    out = {
        "metric": "stereo frame-pairs/sec (1242x375) through pushBack+matchFeatures(2), p_matched bit-exact",
        "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        # untimed calls in front of the W warm-up steps (process start-up) and the slowest call among start-up + warm-up
        "python_gc": "off inside the timed legs (as timeit does), collections between them: a full collection is a 55-65 ms pause of the harness",
        "startup_calls": args.startup, "startup_worst_ms": max(startup_ms) if startup_ms else None,
        "config": {"workload": f"KITTI-shaped synthetic stereo sequence 1242x375, {nf} frames per GPU, quad matching, "
                               "default parameters, replayed Tr_delta feedback, look-ahead C-ABI entry point "
                               f"vsm_sequence_run (chunks of {chunk} frames)",
                   "frames_per_step": nf, "sequences": world, "inputs": "resident in HBM",
                   "results": "when the timed call returns every frame's final list is in host memory as 48-byte p_match records "
                              "(viso/matcher.h:86-100), as Matcher::matchFeatures leaves p_matched_2: the refined lists cross PCIe by DMA as they "
                              "are, the device sends one survivor bit per match behind them, the host pool closes the gaps in "
                              "place - all inside the timed call; vsm_sequence_get_matches copies from there"},
        "per_frame_api": {"value": round(per_frame_value, 3) if per_frame_value else None, "unit": "frame-pairs/s", "caller_thread_on_cpus": caller_cpus,
                          "what": "same sequence through vsm_push_back_device + vsm_match per frame (drop-in "
                                  "Matcher::pushBack/matchFeatures path)"},
        "vo_process_api": {"value": round(vo_value, 3) if vo_value else None, "unit": "frames/s",
                           "after_the_first_frames": {"value": round(vo_steady[0], 3) if vo_steady[0] else None, "skipped": VO_SKIP,
                                                      "what": "rank 0's rate over the frames behind the object's first eight (they set the new handle's ring up)"},
                           "tr_delta_trail_bit_exact_vs_reference": vo_ok,
                           "what": "vsm_vo_stereo_process_device per frame: pushBack + matchFeatures(2, live Tr_delta) + "
                                   "bucketFeatures + RANSAC/Gauss-Newton egomotion (VisualOdometryStereo::process)"},
        "vo_multi_sequence": multi,
        "vo_mono_egomotion": mono,
        "verified_bit_exact_vs_reference_hashes": verified,
        "roofline": roof,
        "roofline_by_kernel": {k: {"achieved_GBps": r["achieved"], "frac": r["frac"], "avg_launch_us": r["avg_launch_us"],
                                   "alone_frac": (r.get("alone") or {}).get("frac"), "alone_us": (r.get("alone") or {}).get("avg_launch_us")}
                               for k, r in roof_all.items()},
        "cpu_baseline": cpu,
        "kernel_ms_per_frame": {k: round(v[0] / nf, 5) for k, v in stats.items() if v[1]},
        "kernel_avg_launch_us": {k: round(v[0] / v[1] * 1e3, 2) for k, v in stats.items() if v[1]},
        "sequence_timings_us": seq_t,
        # what the main stream's kernels alone would sustain: frames / their summed HIP-event durations in the profiled pass
        "gpu_phases_only_frame_pairs_per_s": round(nf / (sum(v[0] for v in stats.values()) * 1e-3), 1),
        "host_threads": int(os.environ["VSM_HOST_THREADS"]),
        "ranks": ranks_info,
        "host_threads_on": NEAR_GPU or "wherever the scheduler puts them",
        "host_cpus": {"os_cpu_count": os.cpu_count(), "cgroup_quota": ncpu, "model": cpu_model()},
        "lookahead_form": {2: "GPU-resident (lists stay in HBM; host only runs Triangle's vertex sort)",
                           1: "host-shared (prior statistics and the top of the exact Delaunay on the host pool)"}.get(lookahead_form),
        "lookahead_host_inputs": {"value": round(host_in_value, 3) if host_in_value else None, "unit": "frame-pairs/s",
                                  "of_resident": round(host_in_value / value, 3) if host_in_value else None,
                                  "bit_exact_vs_reference_hashes": host_in_verified if host_in_value else None,
                                  "calls_ms_rank0": host_in_calls if host_in_value else None,
                                  "what": "the same look-ahead call fed from pageable host memory (on_device = 0): PCIe inclusive; the frames cross "
                                          "in pieces of 20 (pool gathers into pinned memory, DMA on a stream of its own) beside the GPU's work; 6 calls",
                                  "page_locked_by_the_caller": {"value": round(host_pin_value, 3) if host_pin_value else None,
                                                                "of_resident": round(host_pin_value / (value / world), 3) if host_pin_value else None,
                                                                "bit_exact_vs_reference_hashes": host_pin_verified if host_pin_value else None,
                                                                "what": "the same arrays registered once (vsm_host_register) + option seq_host_pinned: DMA straight "
                                                                        "out of the caller's memory, no gather pass; rank 0's own rate"}},
        "secondary_configs": secondary,
        "verified_ranks": n_verified,
        "step_ms_rank0": step_ms,
        "match_timings_us_last_frame": m.timings(),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def multi_sequence_leg(vm, synth, torch, dev, W, H):
    """K independent stereo sequences in lock-step with LIVE Tr_delta feedback (vsm_multi_process = VisualOdometryStereo::
    process for the next pair of every sequence): config 4's eight seeds, repeated to K sequences; aggregate frames/s over
    all sequences, and - for the frames the fixture covers - every sequence's Tr_delta trail against the reference's run of
    that sequence in a process of its own."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_multi_8procs_32f.npz"))
    seeds = [int(x) for x in g["seeds"]]
    nfix, nf = int(g["n_frames"]), 48
    canv = [synth.canvas(sd, W, H) for sd in seeds]
    base = np.stack([np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for cv in canv]) for f in range(nf)])  # [F, 8, 2, H, W]
    base_d = torch.from_numpy(base).to(dev)
    out = {}
    for K in (8, 32, 64):
        idx = torch.arange(K, device=dev) % len(seeds)
        frames = base_d[:, idx]                           # [F, K, 2, H, W], resident in HBM
        left, right = frames[:, :, 0].contiguous(), frames[:, :, 1].contiguous()
        best, exact = None, True
        for rep in range(2):
            gc.collect()
            vo = vm.MultiVisualOdometryStereo(K, *[float(x) for x in g["intr"]])
            torch.cuda.synchronize()
            t = time.perf_counter()
            stamps = []
            for f in range(nf):
                vo.process(left[f], right[f])
                stamps.append(time.perf_counter())
                if rep == 0 and f < nfix:               # (rep 0 is the checked warm-up pass, rep 1 the timed one)
                    for k in range(K):
                        exact = exact and vo.get_motion(k).tobytes() == g[f"s{seeds[k % len(seeds)]}_tr_out"][f].tobytes()
            dt = time.perf_counter() - t
            tm = vo.timings()
            vo.close()
            if rep == 1:
                best = dt
                steady = (nf - 8) / (stamps[-1] - stamps[7])   # steps per second behind the new object's first eight
        out[f"K{K}"] = {"value": round(K * nf / best, 1), "unit": "frames/s (all sequences)", "ms_per_step": round(best / nf * 1e3, 3),
                        "after_the_first_frames": {"value": round(K * steady, 1), "ms_per_step": round(1e3 / steady, 3), "skipped": 8},
                        "tr_delta_trails_bit_exact_vs_reference": bool(exact), "last_step_us": {k: round(v, 1) for k, v in tm.items()}}
    out["what"] = ("vsm_multi_process: pushBack + matchFeatures(2, live Tr_delta) + bucketFeatures + egomotion of K sequences per call, "
                   "one launch per kernel over all K pairs; 48 frames per sequence, images resident in HBM")
    return out


def secondary_configs(vm, synth, torch, dev):
    """BASELINE.json configs[2] (640x480 mono, flow matching) and configs[4] (2048x1024 stereo at ~20 k dense features per
    image, and the denser 40 k variant) on this GPU: throughput plus parity against the reference's committed hashes."""
    import hashlib
    out = {}
    gdir = os.path.join(ROOT, "tests", "golden")

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    # configs[2]: mono input, flow matching - per frame and through the look-ahead entry point
    g = np.load(os.path.join(gdir, "cfg3_640x480_mono.npz"))
    w, h, nfx = int(g["w"]), int(g["h"]), 200
    seq = synth.mono_sequence(int(g["seed"]), w, h, nfx, blur=int(g["blur"]))
    fr = torch.from_numpy(np.stack(seq)).to(dev)
    m = vm.Matcher()
    ok = True
    for f in range(int(g["n_frames"])):
        m.push_back(fr[f])
        m.match_features(0)
        fin = m.get_matches()
        ok = ok and len(fin) == int(g["counts"][f][-1]) and sha(fin) == str(g["hashes"][f][-1])
    m.close()
    m = vm.Matcher()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for f in range(nfx):
        m.push_back(fr[f])
        m.match_features(0)
    dt = time.perf_counter() - t
    m.close()
    m = vm.Matcher()
    got = m.run_sequence(fr[:int(g["n_frames"])], None, 0)
    ok_la = all(len(got[f]) == int(g["counts"][f][-1]) and sha(got[f]) == str(g["hashes"][f][-1]) for f in range(int(g["n_frames"])))
    for _ in range(3):
        m.run_sequence(fr, None, 0, fetch=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        m.run_sequence(fr, None, 0, fetch=False)
    dt_la = (time.perf_counter() - t) / 10
    form = m.sequence_path()
    m.close()
    out["cfg3_640x480_mono_flow"] = {"value": round(nfx / dt, 1), "unit": "frames/s", "api": "per frame (vsm_push_back_device + vsm_match(0))",
                                     "bit_exact_vs_reference_hashes": bool(ok),
                                     "lookahead": {"value": round(nfx / dt_la, 1), "unit": "frames/s", "form": form, "frames": nfx,
                                                   "bit_exact_vs_reference_hashes": bool(ok_la)}}
    # configs[4]
    for name, label in (("cfg5_2048x1024_quad_20k", "cfg5_2048x1024_20k_dense"), ("cfg5_2048x1024_quad", "cfg5_2048x1024_40k_dense")):
        g = np.load(os.path.join(gdir, name + ".npz"))
        w, h, gn = int(g["w"]), int(g["h"]), int(g["n_frames"])
        nfx = 120
        seq = synth.stereo_sequence(int(g["seed"]), w, h, nfx, blur=int(g["blur"]))
        L = torch.from_numpy(np.stack([l for l, _ in seq])).to(dev)
        R = torch.from_numpy(np.stack([r for _, r in seq])).to(dev)
        m = vm.Matcher()
        got = m.run_sequence(L[:gn], R[:gn], 2)
        ok = all(len(got[f]) == int(g["counts"][f][-1]) and sha(got[f]) == str(g["hashes"][f][-1]) for f in range(gn))
        m.run_sequence(L, R, 2, fetch=False)   # (the library's own chunking, as for the headline)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            m.run_sequence(L, R, 2, fetch=False)
        dt = (time.perf_counter() - t) / 4
        form = m.sequence_path()
        chunk5 = int(m.sequence_timings()["chunk"])
        # per-kernel HIP-event durations of one more call, priced like the headline's table (work counters of this
        # configuration: tests/golden/cfg5_work_counters.json, from the oracle - make_cfg5_work_counters.py)
        m.set_profiling(True)
        m.run_sequence(L, R, 2, fetch=False)
        torch.cuda.synchronize()
        kst = m.kernel_stats()
        m.set_profiling(False)
        by_kernel = cfg_kernel_table(kst, nfx, w, h, json.load(open(os.path.join(gdir, "cfg5_work_counters.json"))).get(name))
        for f in range(8):            # (per-frame API: warm-up first - the first frames size the arena and the host pool's buffers)
            m.push_back(L[f], R[f])
            m.match_features(2)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for f in range(8, 56):
            m.push_back(L[f], R[f])
            m.match_features(2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        m.close()
        out[label] = {"lookahead": {"value": round(nfx / dt, 1), "unit": "frame-pairs/s", "form": form, "frames": nfx, "chunk": chunk5},
                      "per_frame_api": {"value": round(48 / dt2, 1), "unit": "frame-pairs/s", "frames": 48, "after_warmup_frames": 8},
                      "roofline_by_kernel": by_kernel,
                      "dense_features_per_image": int(g["counts"][0][1]), "matches_per_pair": int(g["counts"][1][-1]),
                      "blur": int(g["blur"]), "bit_exact_vs_reference_hashes": bool(ok)}
    return out


def cfg_kernel_table(kst, nf, w, h, work):
    """per-kernel table of a secondary configuration: average launch, launches per call, algorithmic bytes per launch and
    the fraction of the HBM peak they amount to (same formulas as the headline: algorithmic_bytes())"""
    P, Ph, Hh = w + 15 - (w - 1) % 16, (w // 2) + 15 - ((w // 2) - 1) % 16, h // 2
    total = sum(ms for k, (ms, n) in kst.items() if n and k != "k_export_list")
    tab = {}
    for k, (ms, nl) in kst.items():
        if not nl:
            continue
        row = {"avg_launch_us": round(ms / nl * 1e3, 2), "launches": int(nl), "share_of_gpu_kernel_time": round(ms / max(total, 1e-9), 3)}
        c2 = dict(P=P, Ph=Ph, Hh=Hh, W=w, H=h, imgs=2.0 * nf / nl)
        ab = None
        if work is not None:
            pairs = (nf - 1) / nl
            w1 = k.endswith("pass1")
            c2.update(Q=work["Q1" if w1 else "Q2"] * pairs, C=work["C1" if w1 else "C2"] * pairs, S=work["S1" if w1 else "S2"] * pairs,
                      Mraw=work["M1" if w1 else "M"] * pairs, M=work["M"] * pairs, N=work["N"])
            ab = algorithmic_bytes(k, c2)
        elif not k.startswith(("k_match", "k_refine", "k_emit")):
            ab = algorithmic_bytes(k, c2)
        if ab:
            row["algorithmic_bytes_per_launch"] = int(ab)
            row["achieved_GBps"] = round(ab / (ms / nl / 1e3) / 1e9, 1)
            row["frac"] = round(ab / (ms / nl / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
        tab[k] = row
    return tab


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
    allc = None
    try:
        allc = cpu_all_cores(kind, host.shape[2], host.shape[3], intr)
    except Exception as e:  # (a box that cannot spawn that many processes still reports the single-core leg)
        allc = {"error": repr(e)}
    return (dict(value=round(n / dt, 3), unit="frame-pairs/s", cores=1, kind=kind,
                 sample=f"first {n} frames of the same sequence, pushBack+matchFeatures(2), 1 thread",
                 vo_process_frames_per_s=vo_fps, all_cores=allc), work)


def _cpu_worker(args):
    """one process = one independent sequence through the CPU matcher (spawned by cpu_all_cores)"""
    kind, seed, w, h, frames, intr, t_go = args
    sys.path.insert(0, ROOT)
    from oracle import bindings as B
    synth = importlib.import_module(PKG + ".synth")
    cv = synth.canvas(seed, w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(frames)]
    cm = B.CpuMatcher("ref" if kind == "reference" else "oracle")
    cm.set_intrinsics(*intr)
    while time.time() < t_go:      # common start
        time.sleep(0.001)
    t0 = time.time()
    for l, r in fr:
        cm.push_back(l, r)
        cm.match(2)
    t1 = time.time()
    cm.close()
    return t0, t1, frames


def cpu_all_cores(kind, w, h, intr, frames=24):
    """the CPU matcher as one process per core, up to every core this job may use (cgroup
    quota), each on its own sequence; aggregate frame-pairs/s over the common window"""
    import multiprocessing as mp
    quota = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = min(quota, max(1, int(q) // int(per)))
    except Exception:
        pass
    procs = max(1, min(quota, 64))
    ctx = mp.get_context("spawn")
    t_go = time.time() + 6.0 + 0.05 * procs     # (imports and frame synthesis happen before the common start)
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(kind, 1234 + i, w, h, frames, intr, t_go) for i in range(procs)])
    t0 = min(r[0] for r in res)
    t1 = max(r[1] for r in res)
    total = sum(r[2] for r in res)
    return dict(value=round(total / (t1 - t0), 3), unit="frame-pairs/s", processes=procs, nproc=os.cpu_count(), cpu_quota=quota,
                cpu_model=cpu_model(), kind=kind,
                sample=f"{procs} processes x {frames} frames 1242x375 (one sequence each, seeds 1234..), pushBack+matchFeatures(2)")


if __name__ == "__main__":
    main()
