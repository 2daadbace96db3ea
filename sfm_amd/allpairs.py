"""Windowed / all-pairs matching sharded over the GPUs of one node.

The reference matches one frame pair per main-loop iteration
(Source Files/app_solver.cpp:404) and re-matches the current frame against up
to `peTMaxIter` earlier track views (Source Files/tracking.cpp:127-135); both
are instances of "match frame a against frame b" over a list of pairs.  Pairs
are independent, so the list is dealt round-robin over the ranks (one process
per GPU, torch.distributed with the RCCL backend over xGMI; gloo on CPU for the
tests).  There is no collective on the data path; the only exchange is the
final all-gather of the match lists, after which every rank holds every pair's
matches in pair order, independent of the number of ranks.
"""
import numpy as np
import torch
import torch.distributed as dist


def window_pairs(n_frames, window=None):
    """(i, j) with i < j (and j - i <= window when given), in lexicographic order."""
    out = [(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)
           if window is None or j - i <= window]
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


def shard_indices(n_pairs, rank, world):
    """Indices of the pairs rank `rank` owns: round-robin by pair index (balanced to one pair)."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


class HipPairMatcher:
    """match_fn for `match_all_pairs`: the HIP library on this rank's GPU."""

    def __init__(self, ctx, norm, d_desc, rows_per_frame, d, ratio, d_counts=None):
        self.ctx, self.norm, self.d_desc = ctx, norm, d_desc
        self.rows, self.d, self.ratio, self.d_counts = rows_per_frame, d, float(ratio), d_counts
        self.n_frames = d_desc.shape[0]
        self.frame_stride = d_desc.stride(0) * d_desc.element_size()
        self._buf = None

    def __call__(self, pairs_np):
        """Returns (counts [P] int32 tensor, matches [sum, 3] int32 tensor: q, t, distance bits)."""
        dev = self.d_desc.device
        P = len(pairs_np)
        cap = self.rows
        if self._buf is None or self._buf[0].shape[0] < P:
            self._buf = (torch.empty((P, cap, 4), dtype=torch.int32, device=dev),
                         torch.empty(P, dtype=torch.int32, device=dev))
        d_m, d_c = self._buf[0][:P], self._buf[1][:P]
        if P == 0:
            return d_c, torch.empty((0, 3), dtype=torch.int32, device=dev)
        d_pairs = torch.from_numpy(np.ascontiguousarray(pairs_np, dtype=np.int32)).to(dev)
        self.ctx.match_pairs_dev(self.norm, self.d_desc, self.n_frames, self.frame_stride, self.rows,
                                 self.d_counts, self.d, d_pairs, P, self.ratio, d_m, cap, d_c)
        return d_c, compact_matches(d_m, d_c)


def compact_matches(d_matches, d_counts):
    """[P, cap, 4] int32 (cv::DMatch layout) + counts -> [sum(counts), 3] (q, t, distance bits)."""
    P, cap, _ = d_matches.shape
    keep = torch.arange(cap, device=d_matches.device)[None, :] < d_counts[:, None]
    return d_matches[keep][:, [0, 1, 3]].contiguous()


class PaddedGather:
    """All-gather of fixed-capacity match buffers with no host synchronisation.

    Every rank contributes its [P, cap, 4] int32 match buffer (cv::DMatch layout) and its [P]
    counts as they are; `start` only enqueues the two collectives (asynchronously, so kernels
    launched afterwards overlap them) and `wait` makes the current stream wait for them.  The
    gathered tensors are [world, P, cap, 4] and [world, P]: rank r's pair p is at [r, p].
    Meant for per-frame-pair use, where P is small and a compaction would cost a host round
    trip per step; windowed matching over many pairs uses `match_all_pairs`.
    """

    def __init__(self, P, cap, world, device, group=None):
        self.world, self.group = world, group
        self.matches = torch.empty((world, P, cap, 4), dtype=torch.int32, device=device)
        self.counts = torch.empty((world, P), dtype=torch.int32, device=device)
        self._work = ()

    def start(self, d_matches, d_counts):
        self._work = (
            dist.all_gather_into_tensor(self.counts.view(-1), d_counts.view(-1), group=self.group,
                                        async_op=True),
            dist.all_gather_into_tensor(self.matches.view(-1), d_matches.view(-1), group=self.group,
                                        async_op=True),
        )

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = ()


def match_all_pairs(pairs, match_fn, rank=0, world=1, group=None):
    """Runs match_fn on this rank's share of `pairs` and gathers every rank's result.

    match_fn(pairs_subset) -> (counts [p] int32, matches [sum(counts), 3] int32) on the
    device the process group works on.  Returns (counts [P] int64 numpy, offsets [P+1]
    int64 numpy, matches [total, 3] int32 tensor): pair k's matches are
    matches[offsets[k]:offsets[k+1]], in the order the matcher produced them.
    """
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    P = len(pairs)
    mine = shard_indices(P, rank, world)
    counts, payload = match_fn(pairs[mine])
    counts = counts.to(torch.int32)
    if world == 1:
        c = counts.cpu().numpy().astype(np.int64)
        off = np.concatenate([[0], np.cumsum(c)])
        return c, off, payload

    dev = payload.device
    per_rank = (P + world - 1) // world
    # 1. counts: every rank sends per_rank entries (-1 pads the short ranks)
    send_c = torch.full((per_rank,), -1, dtype=torch.int32, device=dev)
    send_c[: len(mine)] = counts
    all_c = [torch.empty_like(send_c) for _ in range(world)]
    dist.all_gather(all_c, send_c, group=group)
    all_c = torch.stack(all_c).cpu().numpy().astype(np.int64)  # [world, per_rank]
    totals = np.where(all_c > 0, all_c, 0).sum(axis=1)
    # 2. payload, padded to the largest rank total
    width = int(totals.max()) if totals.size else 0
    send_p = torch.zeros((max(width, 1), 3), dtype=torch.int32, device=dev)
    send_p[: payload.shape[0]] = payload
    all_p = [torch.empty_like(send_p) for _ in range(world)]
    dist.all_gather(all_p, send_p, group=group)
    # 3. reassemble in pair order: pair k lives on rank k % world at slot k // world
    k = np.arange(P)
    r_of, s_of = k % world, k // world
    counts_out = all_c[r_of, s_of]
    offsets = np.concatenate([[0], np.cumsum(counts_out)])
    total = int(offsets[-1])
    if total == 0:
        return counts_out, offsets, torch.empty((0, 3), dtype=torch.int32, device=dev)
    valid = np.where(all_c > 0, all_c, 0)
    rank_base = np.concatenate([[0], np.cumsum(totals)])[:world]          # start of rank r in `cat`
    slot_off = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(valid, axis=1)], axis=1)
    src_start = rank_base[r_of] + slot_off[r_of, s_of]
    gather = np.repeat(src_start - offsets[:-1], counts_out) + np.arange(total)
    cat = torch.cat([all_p[r][: int(totals[r])] for r in range(world)])
    matches = cat[torch.from_numpy(gather).to(dev)]
    return counts_out, offsets, matches
