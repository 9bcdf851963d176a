"""Row tiling across GPUs + framebuffer assembly (north_star: "row-tiled across the 8 GPUs of one node
with an RCCL gather over xGMI").  One process per GPU under torch.distributed; the unit of
partition is the image row, the reference's own unit of parallelism (smallpt.cpp:317,736).

The RNG is keyed by the GLOBAL pixel index, so the assembled image is bit-identical for any world
size.  The only exchange step moves row_count*w*3 floats per rank to rank 0 (no reduction): every other
rank sends its band, the root receives each band STRAIGHT INTO ITS ROW-SLICE of one framebuffer
(batched point-to-point ops = grouped ncclSend/ncclRecv with the nccl backend: 7 xGMI links into the
root in parallel, no ring, no staging copies, no concatenation).  The root renders its own band in
place.  The single-process C++ counterpart is csrc/spt_multi.cpp (include/smallpt_mi355x_multi.h).
"""
import torch
import torch.distributed as dist


def row_band(h, world_size, rank):
    """Contiguous band [begin, begin+count) of rank `rank`: rows split as evenly as possible, the first
    (h % world_size) ranks get one extra row.  Bands are in rank order = row order."""
    base, extra = divmod(h, world_size)
    count = base + (1 if rank < extra else 0)
    begin = rank * base + min(rank, extra)
    return begin, count


def interleaved_rows(h, block_rows, world_size, rank):
    """Image rows of rank `rank` when the rows are dealt out round-robin in blocks of `block_rows` rows (block t belongs
    to rank t % world_size), ascending.  Contiguous bands of a Cornell-like image differ by up to 1.34x in cost (floor and
    spheres below, ceiling above: tools/probe_band_balance.py); interleaved blocks balance the ranks."""
    rows = []
    for start in range(rank * block_rows, h, world_size * block_rows):
        rows.extend(range(start, min(start + block_rows, h)))
    return rows


class FrameAssembler:
    """Owns the destination memory of the exchange: on `dst` the whole (h, w, 3) framebuffer, elsewhere this rank's
    rows.  ``band`` is the tensor a rank renders into; ``gather()`` runs the exchange and returns the framebuffer on dst,
    None elsewhere.

    interleave = 0: contiguous bands (row_band); on dst ``band`` is a view of its rows inside the framebuffer and the
    other bands are received straight into their row slices.
    interleave = B > 0: rows dealt out round-robin in blocks of B rows (interleaved_rows); every rank's packed rows are
    received into a staging tensor on dst and scattered to their rows with one index_copy per rank."""

    def __init__(self, w, h, device="cpu", group=None, dst=0, dtype=torch.float32, interleave=0):
        self.w, self.h, self.group, self.dst, self.interleave = w, h, group, dst, int(interleave)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.interleave and self.world > 1:
            self.rows = [interleaved_rows(h, self.interleave, self.world, r) for r in range(self.world)]
            self.begin, self.count = None, len(self.rows[self.rank])
            self.band = torch.empty((self.count, w, 3), dtype=dtype, device=device)
            if self.rank == dst:
                self.frame = torch.empty((h, w, 3), dtype=dtype, device=device)
                self._index = [torch.tensor(r, dtype=torch.long, device=device) for r in self.rows]
                self._staging = [self.band if r == dst else torch.empty((len(self.rows[r]), w, 3), dtype=dtype, device=device)
                                 for r in range(self.world)]
            else:
                self.frame = None
            return
        self.interleave = 0
        self.begin, self.count = row_band(h, self.world, self.rank)
        if self.rank == dst:
            self.frame = torch.empty((h, w, 3), dtype=dtype, device=device)
            self.band = self.frame[self.begin:self.begin + self.count]
        else:
            self.frame = None
            self.band = torch.empty((self.count, w, 3), dtype=dtype, device=device)

    def all_ok(self, ok=True):
        """Agreement on whether every rank's render succeeded (one 1-element all_reduce, the only collective besides the
        point-to-point exchange).  A rank whose render failed must not leave its peers waiting in irecv / isend for rows that
        will never come: callers pass their local status, and when any rank reports failure EVERY rank skips the exchange."""
        if self.world == 1:
            return bool(ok)
        dev = self.band.device if self.band.device.type == "cuda" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def gather(self, ok=None):
        """Runs the exchange.  ok = this rank's render status (True / False) makes the ranks agree first (all_ok) and raises
        RuntimeError on every rank if any of them failed; ok = None skips the agreement (the caller vouches for all ranks)."""
        if ok is not None and not self.all_ok(ok):
            raise RuntimeError("FrameAssembler.gather: the render failed on " + ("this rank" if not ok else "another rank") + "; exchange skipped on every rank")
        if self.world == 1:
            return self.frame
        ops = []
        if self.rank == self.dst:
            for r in range(self.world):
                if r == self.dst:
                    continue
                if self.interleave:
                    if len(self.rows[r]):
                        ops.append(dist.P2POp(dist.irecv, self._staging[r], r, self.group))
                else:
                    b, c = row_band(self.h, self.world, r)
                    if c:
                        ops.append(dist.P2POp(dist.irecv, self.frame[b:b + c], r, self.group))
        elif self.count:
            ops.append(dist.P2POp(dist.isend, self.band, self.dst, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.interleave and self.rank == self.dst:
            for r in range(self.world):
                if len(self.rows[r]):
                    self.frame.index_copy_(0, self._index[r], self._staging[r])
        return self.frame


def gather_rows(band, w, h, group=None, dst=0):
    """Assembles the (h, w, 3) framebuffer on rank `dst` from every rank's (count, w, 3) band (a convenience over
    FrameAssembler for callers that already hold their band: costs one device copy of the root's own rows)."""
    fa = FrameAssembler(w, h, device=band.device, group=group, dst=dst, dtype=band.dtype)
    assert band.shape == (fa.count, w, 3), (band.shape, fa.count, w)
    if fa.world == 1:
        return band
    if fa.rank == dst:
        fa.band.copy_(band)
    else:
        fa.band = band.contiguous()
    return fa.gather()


def render_distributed(render_band, w, h, group=None, dst=0):
    """render_band(row_begin, row_count) -> (row_count, w, 3) tensor of this rank's rows (the HIP path on
    a GPU rank).  Returns the assembled image on dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    begin, count = row_band(h, world, rank)
    band = render_band(begin, count)
    return gather_rows(band, w, h, group=group, dst=dst)
