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

Two more pieces for the same setting.  `replicate_frames`: every rank holds the descriptors
of the frames f with f % world == rank (it produced or received those) and one all-gather
gives every rank the whole stack -- the one sizeable payload of the configuration.  And the
second partition, for windows with fewer pairs than ranks (the reference's own window is a
handful of earlier views against the current frame): `match_pair_slabs` splits ONE pair by
query rows -- rank r searches its slab of prev rows against all of curr and its slab of curr
rows against all of prev, the [rows][2] neighbour arrays are all-gathered, and the ratio test
and cross-check run replicated on the gathered arrays.  `match_all_pairs` takes that route by
itself when there are fewer pairs than ranks and a slab matcher was given.
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


MATCH16_DTYPE = np.dtype([("queryIdx", "<u2"), ("trainIdx", "<u2"), ("distance", "<u2")])


class HipPairMatcher:
    """match_fn for `match_all_pairs`: the HIP library on this rank's GPU.

    Matches come back in the library's compact exchange format (sfm_match16: prev row, curr
    row, Hamming distance as three uint16, held here as an int16 tensor [n, 3]) written back to
    back in pair order by sfm_match_pairs_compact_dev; the rank never holds a [pairs, rows]
    cv::DMatch buffer.  The context must enqueue on torch's current stream (the tensors this
    class hands out are consumed by torch and by the collectives on that stream).
    """

    def __init__(self, ctx, norm, d_desc, rows_per_frame, d, ratio, d_counts=None, matches_per_pair=None):
        if ctx.stream is None or ctx.stream != torch.cuda.current_stream(d_desc.device).cuda_stream:
            raise ValueError("HipPairMatcher needs a Context created on torch's current stream "
                             "(capi.Context(dev, stream=torch.cuda.current_stream().cuda_stream))")
        self.ctx, self.norm, self.d_desc = ctx, norm, d_desc
        self.rows, self.d, self.ratio, self.d_counts = rows_per_frame, d, float(ratio), d_counts
        self.n_frames = d_desc.shape[0]
        self.frame_stride = d_desc.stride(0) * d_desc.element_size()
        # first guess of the output size; grown (and the call repeated) when a window needs more
        self.per_pair = int(matches_per_pair) if matches_per_pair else max(rows_per_frame // 4, 16)
        self._out = self._counts = self._offsets = self._pairs = None

    def _buffers(self, P, records):
        dev = self.d_desc.device
        if self._out is None or self._out.shape[0] < records:
            self._out = torch.empty((records, 3), dtype=torch.int16, device=dev)
        if self._counts is None or self._counts.shape[0] < P:
            self._counts = torch.empty(P, dtype=torch.int32, device=dev)
            self._offsets = torch.empty(P + 1, dtype=torch.int64, device=dev)
            self._pairs = torch.empty((P, 2), dtype=torch.int32, device=dev)

    def __call__(self, pairs_np):
        """Returns (counts [P] int32 tensor, matches [sum(counts), 3] int16 tensor holding
        uint16 (q, t, distance) records).  Both are views into buffers this object reuses: they
        are valid until its next call (match_all_pairs copies what it hands on)."""
        dev = self.d_desc.device
        P = len(pairs_np)
        if P == 0:
            return (torch.empty(0, dtype=torch.int32, device=dev),
                    torch.empty((0, 3), dtype=torch.int16, device=dev))
        self._buffers(P, max(P * self.per_pair, 1024))
        self._pairs[:P].copy_(torch.from_numpy(np.ascontiguousarray(pairs_np, dtype=np.int32)))
        while True:
            self.ctx.match_pairs_compact_dev(self.norm, self.d_desc, self.n_frames, self.frame_stride,
                                             self.rows, self.d_counts, self.d, self._pairs, P, self.ratio,
                                             self._out, self._out.shape[0], self._counts, self._offsets)
            total = int(self._offsets[P].item())       # one host read per window
            if total <= self._out.shape[0]:
                return self._counts[:P], self._out[:total]
            self.per_pair = total // P + 1              # overflow: grow and repeat
            self._buffers(P, int(total * 1.05) + 1024)

    def gather_segments(self, src, src_off, dst_off, lens, total):
        """dst[dst_off[s] : dst_off[s] + lens[s]] = src[src_off[s] : ...] on the device."""
        dev = src.device
        dst = torch.empty((total, src.shape[1]), dtype=src.dtype, device=dev)
        if total:
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
            self.ctx.gather_segments_dev(src, t(src_off, torch.int64), t(dst_off, torch.int64),
                                         t(lens, torch.int32), len(lens), src.shape[1] * src.element_size(),
                                         dst)
        return dst


def gather_segments_host(src, src_off, dst_off, lens, total):
    """CPU tensors: the same reassembly with an index built on the host."""
    if total == 0:
        return src[:0].clone()
    idx = np.repeat(np.asarray(src_off) - np.asarray(dst_off), lens) + np.arange(total)
    return src[torch.from_numpy(idx)]


class PaddedGather:
    """All-gather of fixed-capacity match buffers with no host synchronisation.

    Every rank contributes a match buffer of `payload_shape` and its [P] counts as they are;
    `start` only enqueues the two collectives (asynchronously, so kernels launched afterwards
    overlap them) and `wait` makes the current stream wait for them.  The gathered tensors are
    [world, *payload_shape] and [world, P].  The payload is either the [P, cap, 4] int32 buffer
    of cv::DMatch records (rank r's pair p at [r, p]) or, at 6 instead of 16 bytes per match, the
    [P * cap, 3] int16 buffer the compact matcher fills back to back (rank r's pair p starts at
    the sum of r's counts before p; only that prefix of a rank's slice is meaningful).
    Meant for per-frame-pair use, where P is small and sizing the exchange by the counts would
    cost a host round trip per step; windowed matching over many pairs uses `match_all_pairs`.
    """

    def __init__(self, P, cap, world, device, group=None, compact=False):
        self.world, self.group = world, group
        shape, dtype = ((P * cap, 3), torch.int16) if compact else ((P, cap, 4), torch.int32)
        self.matches = torch.empty((world,) + shape, dtype=dtype, device=device)
        self.counts = torch.empty((world, P), dtype=torch.int32, device=device)
        self._work = ()

    def start(self, d_matches, d_counts):
        self._work = (
            dist.all_gather_into_tensor(self.counts.view(-1), d_counts.view(-1), group=self.group,
                                        async_op=True),
            # as bytes: the process group takes no 16-bit integers
            dist.all_gather_into_tensor(self.matches.view(-1).view(torch.uint8),
                                        d_matches.view(-1).view(torch.uint8), group=self.group, async_op=True),
        )

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = ()


def reassemble(all_c, all_p, width, n_pairs, world, gather):
    """Pair order out of the gathered per-rank results.

    all_c: [world, per_rank] int64 numpy, rank r's per-pair counts (-1 pads); all_p: the ranks'
    payloads one after another, each `width` records long; pair i lives on rank i % world at slot
    i // world.  Returns (counts [P], offsets [P+1], matches [total, k])."""
    valid = np.where(all_c > 0, all_c, 0)
    i = np.arange(n_pairs)
    r_of, s_of = i % world, i // world
    counts_out = all_c[r_of, s_of]
    offsets = np.concatenate([[0], np.cumsum(counts_out)])
    total = int(offsets[-1])
    slot_off = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(valid, axis=1)], axis=1)
    src_start = r_of * width + slot_off[r_of, s_of]
    matches = gather(all_p, src_start.astype(np.int64), offsets[:-1].astype(np.int64),
                     counts_out.astype(np.int32), total)
    return counts_out, offsets, matches


def slab_bounds(n_rows, rank, world):
    """Rows [lo, hi) of rank `rank` when n_rows are cut into `world` slabs of near-equal size."""
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


def replicate_frames(local, n_frames, rank, world, group=None):
    """The whole descriptor stack on every rank from per-rank shares.

    local: [ceil(n_frames / world), rows, d] tensor on the group's device; slot s holds frame
    s * world + rank (slots beyond the last frame are padding).  One all-gather (as bytes),
    then a strided copy into frame order.  Returns (stack [n_frames, rows, d], bytes received
    from the other ranks)."""
    per = (n_frames + world - 1) // world
    assert local.shape[0] == per, (local.shape, per)
    if world == 1:
        return local[:n_frames], 0
    local = local.contiguous()
    gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered.view(torch.uint8).view(-1), local.view(torch.uint8).view(-1), group=group)
    # gathered[r, s] is frame s * world + r
    stack = gathered.transpose(0, 1).reshape((per * world,) + tuple(local.shape[1:]))[:n_frames].contiguous()
    return stack, (world - 1) * local.numel() * local.element_size()


class HipSlabMatcher:
    """The two callables `match_pair_slabs` needs, on this rank's GPU: 2-NN of a slab of query
    rows (sfm_knn2_dev) and the replicated ratio test + cross-check (sfm_ratio_cross_dev).
    d_desc: [frames, rows, d] descriptors, every frame with `rows` rows."""

    def __init__(self, ctx, norm, d_desc, ratio):
        if ctx.stream is None or ctx.stream != torch.cuda.current_stream(d_desc.device).cuda_stream:
            raise ValueError("HipSlabMatcher needs a Context created on torch's current stream")
        self.ctx, self.norm, self.d_desc, self.ratio = ctx, norm, d_desc, float(ratio)
        self.rows, self.d = d_desc.shape[1], d_desc.shape[2]
        self.row_bytes = d_desc.stride(1) * d_desc.element_size()

    def n_rows(self, frame):
        return self.rows

    def knn_slab(self, q_frame, t_frame, lo, hi):
        dev = self.d_desc.device
        idx = torch.empty((hi - lo, 2), dtype=torch.int32, device=dev)
        dst = torch.empty((hi - lo, 2), dtype=torch.float32, device=dev)
        if hi > lo:
            self.ctx.knn2_dev(self.norm, self.d_desc[q_frame, lo:hi], hi - lo, self.row_bytes, self.d_desc[t_frame],
                              self.rows, self.row_bytes, self.d, idx, dst)
        return idx, dst

    def cross(self, idx_f, dist_f, idx_b, dist_b):
        dev = self.d_desc.device
        n_prev, n_curr = idx_f.shape[0], idx_b.shape[0]
        cap = min(n_prev, n_curr)
        out = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ctx.ratio_cross_dev(idx_f, dist_f, n_prev, idx_b, dist_b, n_curr, self.ratio, out, cap, cnt)
        n = int(cnt.item())
        m = out[:n]
        # the exchange format of the pair partition: (prev row, curr row, distance) as 16-bit words
        rec = torch.stack([m[:, 0], m[:, 1], m[:, 3].view(torch.float32).to(torch.int32)], 1).to(torch.int16)
        return n, rec


def _gather_rows(part, n_rows, rank, world, group):
    """[n_rows, k] from every rank's slab [hi - lo, k] (slabs as slab_bounds cuts them)."""
    if world == 1:
        return part
    width = max(slab_bounds(n_rows, r, world)[1] - slab_bounds(n_rows, r, world)[0] for r in range(world))
    send = torch.zeros((max(width, 1), part.shape[1]), dtype=part.dtype, device=part.device)
    send[: part.shape[0]] = part
    recv = torch.empty((world,) + tuple(send.shape), dtype=part.dtype, device=part.device)
    dist.all_gather_into_tensor(recv.view(torch.uint8).view(-1), send.view(torch.uint8).view(-1), group=group)
    return torch.cat([recv[r, : slab_bounds(n_rows, r, world)[1] - slab_bounds(n_rows, r, world)[0]]
                      for r in range(world)])


def match_pair_slabs(pair, slab, rank=0, world=1, group=None):
    """ONE frame pair split by query rows over the ranks; every rank returns the whole result.

    slab: an object with n_rows(frame), knn_slab(query frame, train frame, lo, hi) -> (idx
    [hi - lo, 2] int32, dist [hi - lo, 2] float32) and cross(idx_f, dist_f, idx_b, dist_b) ->
    (count, matches [count, k]).  Four all-gathers of [rows][2] arrays (8 bytes per row and
    direction each way); no reduction across ranks is needed, the two nearest rows of a query
    are found by the rank that owns the query."""
    a, b = int(pair[0]), int(pair[1])
    n_prev, n_curr = slab.n_rows(a), slab.n_rows(b)
    lo, hi = slab_bounds(n_prev, rank, world)
    idx_f, dist_f = slab.knn_slab(a, b, lo, hi)
    lo, hi = slab_bounds(n_curr, rank, world)
    idx_b, dist_b = slab.knn_slab(b, a, lo, hi)
    idx_f = _gather_rows(idx_f, n_prev, rank, world, group)
    dist_f = _gather_rows(dist_f, n_prev, rank, world, group)
    idx_b = _gather_rows(idx_b, n_curr, rank, world, group)
    dist_b = _gather_rows(dist_b, n_curr, rank, world, group)
    return slab.cross(idx_f, dist_f, idx_b, dist_b)


def match_all_pairs(pairs, match_fn, rank=0, world=1, group=None, always_exchange=False, shard="auto",
                    slab=None):
    """Runs match_fn on this rank's share of `pairs` and gathers every rank's result.

    match_fn(pairs_subset) -> (counts [p] int32, matches [sum(counts), k]) on the device the
    process group works on (k columns of any one dtype: three int16 holding the library's
    uint16 records for HipPairMatcher).  Returns (counts [P] int64 numpy, offsets [P+1] int64
    numpy, matches [total, k] tensor): pair i's matches are matches[offsets[i]:offsets[i+1]],
    in the order the matcher produced them, whatever the number of ranks.

    Exchange: one all-gather of the per-pair counts (int32, padded to the largest share), one
    host read of them, one all-gather of the payloads padded to the largest rank total (bytes),
    then a segment copy into pair order.  With one rank there is nothing to exchange unless
    always_exchange asks for the collectives anyway (a one-rank group: how a single GPU
    exercises this path).

    shard: 'pair' deals whole pairs round-robin (above); 'slab' splits every pair by query rows
    over all ranks (`match_pair_slabs`, needs `slab`); 'auto' takes 'slab' when a slab matcher
    is given and there are fewer pairs than ranks, 'pair' otherwise.  The result is the same.
    """
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    P = len(pairs)
    if shard not in ("auto", "pair", "slab"):
        raise ValueError("shard must be 'auto', 'pair' or 'slab'")
    if shard == "slab" or (shard == "auto" and slab is not None and world > 1 and P < world):
        # fewer pairs than ranks: every pair is split by query rows over all ranks
        if slab is None:
            raise ValueError("shard='slab' needs a slab matcher")
        res = [match_pair_slabs(pr, slab, rank, world, group) for pr in pairs]
        c = np.asarray([n for n, _ in res], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
        if res:
            return c, off, torch.cat([m for _, m in res])
        return c, off, None
    mine = shard_indices(P, rank, world)
    counts, payload = match_fn(pairs[mine])
    counts = counts.to(torch.int32)
    if world == 1 and not always_exchange:
        c = counts.cpu().numpy().astype(np.int64)
        off = np.concatenate([[0], np.cumsum(c)])
        # a matcher may hand out views into buffers it reuses on its next call
        return c, off, payload.clone()

    dev = payload.device
    per_rank = (P + world - 1) // world
    # 1. counts: every rank sends per_rank entries (-1 pads the short ranks)
    send_c = torch.full((per_rank,), -1, dtype=torch.int32, device=dev)
    send_c[: len(mine)] = counts
    all_c = torch.empty((world, per_rank), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_c.view(-1), send_c, group=group)
    all_c = all_c.cpu().numpy().astype(np.int64)  # [world, per_rank]
    totals = np.where(all_c > 0, all_c, 0).sum(axis=1)
    # 2. payload as bytes, padded to the largest rank total
    width = max(int(totals.max()) if totals.size else 0, 1)
    k = payload.shape[1]
    send_p = torch.zeros((width, k), dtype=payload.dtype, device=dev)
    send_p[: payload.shape[0]] = payload
    all_p = torch.empty((world * width, k), dtype=payload.dtype, device=dev)
    dist.all_gather_into_tensor(all_p.view(torch.uint8).view(-1), send_p.view(torch.uint8).view(-1), group=group)
    # 3. pair order
    gather = getattr(match_fn, "gather_segments", None) if dev.type != "cpu" else None
    return reassemble(all_c, all_p, width, P, world, gather or gather_segments_host)
