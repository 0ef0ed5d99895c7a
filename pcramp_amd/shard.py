"""Target sharding across the GPUs of one node (one process per GPU, torch.distributed).

Evaluations are independent per target, so the sequence set is cut into `world` contiguous
blocks whose boundaries are multiples of 64 sequences (bitset words never straddle ranks) and
whose sizes are balanced by bases.  Each rank screens its block; the only exchange per pass is
ONE all-gather of the per-pair bitset words (RCCL over xGMI on the GPU box, gloo in the CPU
tests).  Weighted coverage is then re-summed identically on every rank from the gathered bits
(pcr_coverage_from_bits keeps the reference's summation order).  The reference itself has no
target sharding (its MPI mode shards trials, main.cpp:65); this layer is new.
"""
import numpy as np


def shard_ranges(lengths, world):
    """-> list of (lo, hi) sequence index ranges, lo a multiple of 64, balanced by total bases."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.size
    nblk = (n + 63) // 64
    blk_bases = np.add.reduceat(lengths, np.arange(0, n, 64)) if n else np.zeros(0, np.int64)
    cum = np.concatenate([[0], np.cumsum(blk_bases)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        want = total * r / world
        b = int(np.searchsorted(cum, want, side="left"))
        b = max(cuts[-1], min(nblk, b))
        cuts.append(b)
    cuts.append(nblk)
    return [(min(n, cuts[r] * 64), min(n, cuts[r + 1] * 64)) for r in range(world)]


def gather_bitsets(local_words, ranges, group=None):
    """local_words: integer tensor [..., words_local] of this rank's shard (64 targets per word).
    Returns the full-width tensor [..., ceil(n/64)] on every rank, using ONE all_gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    words = [(hi - lo + 63) // 64 for lo, hi in ranges]
    wmax = max(words) if words else 0
    lead = tuple(local_words.shape[:-1])
    pad = torch.zeros(lead + (wmax,), dtype=local_words.dtype, device=local_words.device)
    pad[..., :local_words.shape[-1]] = local_words
    flat_in = pad.contiguous().view(-1)
    flat_out = torch.empty(world * flat_in.numel(), dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(flat_out, flat_in, group=group)
    out = flat_out.view((world,) + lead + (wmax,))
    return torch.cat([out[r][..., :words[r]] for r in range(world)], dim=-1)
