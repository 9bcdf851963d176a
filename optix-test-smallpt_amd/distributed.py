"""Row tiling across GPUs + framebuffer assembly (north_star: "row-tiled across the 8 GPUs of one node
with an RCCL gather over xGMI").  One process per GPU under torch.distributed; the unit of
partition is the image row, the reference's own unit of parallelism (smallpt.cpp:317,736).

The RNG is keyed by the GLOBAL pixel index, so the assembled image is bit-identical for any world
size.  The only exchange step is one gather of row_count*w*3 floats per rank to rank 0 (no
reduction); with the nccl backend that is RCCL send/recv over xGMI, 7 point-to-point links into
the root in parallel.
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


def gather_rows(band, w, h, group=None, dst=0):
    """Assembles the (h, w, 3) framebuffer on rank `dst` from every rank's (count, w, 3) band.
    Returns the full image tensor on dst, None elsewhere.  Uneven bands are padded to the largest
    band so that one collective moves everything."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    begin, count = row_band(h, world, rank)
    assert band.shape == (count, w, 3), (band.shape, count, w)
    if world == 1:
        return band
    max_count = row_band(h, world, 0)[1]
    if count != max_count:
        padded = band.new_zeros((max_count, w, 3))
        padded[:count] = band
    else:
        padded = band.contiguous()
    if rank == dst:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, parts, dst=dst, group=group)
        rows = [parts[r][: row_band(h, world, r)[1]] for r in range(world)]
        return torch.cat(rows, dim=0)
    dist.gather(padded, None, dst=dst, group=group)
    return None


def render_distributed(render_band, w, h, group=None, dst=0):
    """render_band(row_begin, row_count) -> (row_count, w, 3) tensor of this rank's rows (the HIP path on
    a GPU rank).  Returns the assembled image on dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    begin, count = row_band(h, world, rank)
    band = render_band(begin, count)
    return gather_rows(band, w, h, group=group, dst=dst)
