"""Hypothesis sharding across the GPUs of one node.

Hypotheses are independent, so the propose+score stage shards with no data-path
collective: correspondences (32 B/point) are replicated, rank r owns the RNG
counters [first_r, first_r + m_r) of the global batch, and the only exchange is
one all-gather of the per-model int32 inlier scores (4*M bytes in total,
latency-bound on xGMI) after which every rank runs the same selection on the
same array.  One process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests).

The functions here are device-agnostic tensor plumbing; the scoring itself is
the engine's kernel (or, in the CPU tests only, the oracle standing in for it).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.distributed as dist


def shard_counts(total: int, world: int) -> list[int]:
    """Sizes of the `world` contiguous shards of `total` hypotheses (first shards one longer)."""
    base, rem = divmod(int(total), int(world))
    return [base + (1 if r < rem else 0) for r in range(world)]


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """(first, count) of rank's shard inside a global batch of `total` hypotheses."""
    sizes = shard_counts(total, world)
    return sum(sizes[:rank]), sizes[rank]


def batch_first(step: int, world: int, rank: int, per_rank: int) -> int:
    """Weak scaling: every (step, rank) owns a disjoint block of `per_rank` RNG counters."""
    return (int(step) * int(world) + int(rank)) * int(per_rank)


def gather_scores(local: torch.Tensor, world: int, out: torch.Tensor | None = None,
                  sizes: list[int] | None = None) -> torch.Tensor:
    """All-gather the per-model scores of every rank into one flat int32 tensor, rank order.

    Equal shard sizes use all_gather_into_tensor (one RCCL call, no copies); ragged shards
    (strong scaling with world not dividing M) pad to the longest shard and strip the padding."""
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal on a box with fewer GPUs than ranks (tests): gloo moves host memory, so stage through it.
        # On a real node the backend is "nccl" (= RCCL) and the branches below run on the device buffers.
        return gather_scores(local.cpu(), world, None, sizes).to(local.device)
    if sizes is None or len(set(sizes)) == 1:
        if out is None:
            out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    longest = max(sizes)
    padded = torch.full((longest,), -1, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    buf = torch.empty(world * longest, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded)
    return torch.cat([buf[r * longest: r * longest + sizes[r]] for r in range(world)])


def select_best(scores: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Index (into the gathered array) and score of the best-supported model; first maximum wins,
    so every rank picks the same model from the same gathered array."""
    best = torch.argmax(scores)
    return best, scores[best]


def global_model_index(flat_index: int, sizes: list[int]) -> tuple[int, int]:
    """(rank, local index) of an index into the gathered score array."""
    r = 0
    while flat_index >= sizes[r]:
        flat_index -= sizes[r]
        r += 1
    return r, flat_index


# Transport hook of the host class (MultiH::SetSharding, multi-h_amd/host/MultiH.h) and of mh_select_greedy: the engine
# hands over DEVICE pointers (its resident int32 score buffer on the send side); the exchange is torch.distributed's
# all-gather on zero-copy views of them — RCCL when the process group is "nccl".  With "gloo" (tests: several ranks
# share one GPU, which RCCL refuses) the views are staged through host memory.
ALLGATHER_CFUNC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong)


class _DevBytes:
    """Zero-copy uint8 view of device memory for torch (CUDA array interface)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2,
                                         "strides": None}


def make_allgather_hook(world: int, device: torch.device | None = None, group=None):
    """ctypes callback for mhh_set_sharding / MultiH::SetSharding / mh_select_greedy.  `device` = the rank's GPU (the
    pointers the engine passes are device pointers).  device=None treats the pointers as HOST memory: only the CPU
    tests of the rank-ordered concatenation use that.  Keep the returned object alive for as long as the host library
    may call it."""
    stats = {"calls": 0, "bytes": 0}

    def hook(_ctx, send, recv, nbytes):
        try:
            if device is None:
                src = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(send))
                dst = np.ctypeslib.as_array((ctypes.c_uint8 * (nbytes * world)).from_address(recv))
                dist.all_gather_into_tensor(torch.from_numpy(dst), torch.from_numpy(src), group=group)
                stats["calls"] += 1
                stats["bytes"] += nbytes * world
                return 0
            src = torch.as_tensor(_DevBytes(send, nbytes), device=device)
            dst = torch.as_tensor(_DevBytes(recv, nbytes * world), device=device)
            if dist.get_backend(group) == "gloo":
                out = torch.empty(nbytes * world, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, src.cpu(), group=group)
                dst.copy_(out)
            else:
                dist.all_gather_into_tensor(dst, src, group=group)        # RCCL on the engine's own buffers
            torch.cuda.synchronize(device)                                # the engine resumes on its own stream
            stats["calls"] += 1
            stats["bytes"] += nbytes * world
            return 0
        except Exception as exc:  # never let an exception cross the C boundary
            print(f"[multi-h sharding] all-gather failed: {exc!r}", flush=True)
            return 1

    cb = ALLGATHER_CFUNC(hook)
    cb.stats = stats
    return cb
