"""One process per GPU, one independent stereo sequence per process.

The path shards by sequence with no data-path exchange: no image, feature or match ever crosses
GPUs.  The only collectives are a start barrier and the reduction that turns per-rank
(units, seconds) into whole-job throughput: sum(units) / max(seconds).  Backend "nccl" (= RCCL over
xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

BASE_SEED = 1234


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def sequence_seed(rank):
    """rank r owns the sequence generated from seed 1234 + r (BASELINE.json configs[3])"""
    return BASE_SEED + rank


def barrier(dist_mod, device=None):
    if dist_mod is not None and dist_mod.is_initialized():
        if device is not None and device.type == "cuda":
            dist_mod.barrier(device_ids=[device.index])
        else:
            dist_mod.barrier()


def aggregate(dist_mod, torch_mod, units, seconds, device=None, all_ok=True):
    """returns (total_units, max_seconds, all_ranks_ok) -- identical on every rank"""
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size() == 1:
        return float(units), float(seconds), bool(all_ok)
    dev = device if device is not None else torch_mod.device("cpu")
    s = torch_mod.tensor([float(units)], dtype=torch_mod.float64, device=dev)
    t = torch_mod.tensor([float(seconds)], dtype=torch_mod.float64, device=dev)
    o = torch_mod.tensor([1 if all_ok else 0], dtype=torch_mod.int32, device=dev)
    dist_mod.all_reduce(s, op=dist_mod.ReduceOp.SUM)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    dist_mod.all_reduce(o, op=dist_mod.ReduceOp.MIN)
    return float(s.item()), float(t.item()), bool(o.item())


def gather_rows(dist_mod, torch_mod, row, device=None):
    """every rank's row of a few numbers (host threads, chunk plan, its own step time ...) on every rank: a SCALE run then
    explains its own efficiency in the rank-0 line.  A few bytes per rank, once, outside the timed region."""
    row = [float(x) for x in row]
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size() == 1:
        return [row]
    dev = device if device is not None else torch_mod.device("cpu")
    mine = torch_mod.tensor(row, dtype=torch_mod.float64, device=dev)
    out = [torch_mod.zeros_like(mine) for _ in range(dist_mod.get_world_size())]
    dist_mod.all_gather(out, mine)
    return [[float(x) for x in t.tolist()] for t in out]
