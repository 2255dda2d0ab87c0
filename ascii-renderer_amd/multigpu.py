"""Screen-strip partition for multi-GPU runs: one process per GPU, each renders a contiguous band of
output rows from its own replica of the mesh; the ASCII bands are combined with one all-gather per
frame (torch.distributed: RCCL on GPUs, gloo in the CPU tests).  The reference is single-GPU; this
is the build's own scaling layer on top of the C ABI's ar_set_strip."""
import torch
import torch.distributed as dist


def strip_bounds(rows, world, rank):
    """Rows [row0, row1) of rank `rank`: equal-height bands, as the all-gather needs equal sizes."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank %d of %d" % (rank, world))
    if rows % world:
        raise ValueError("rows (%d) must divide evenly over %d ranks" % (rows, world))
    h = rows // world
    return rank * h, (rank + 1) * h


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_strip_tensor(renderer, cols, rows, row0, row1):
    """uint8 tensor [row1-row0, cols] aliasing the renderer's ASCII output rows in HBM (no copy)."""
    base = renderer.device_buffer("ascii")
    if not base:
        raise RuntimeError("render one frame before asking for the device buffer")
    return torch.as_tensor(_DeviceArray(base + row0 * cols, (row1 - row0, cols), "|u1"), device="cuda")


def all_gather_strips(local_strip, out=None):
    """Concatenates every rank's band (rank order == row order) into the full grid on every rank."""
    world = dist.get_world_size()
    h, cols = local_strip.shape
    if out is None:
        out = torch.empty((h * world, cols), dtype=local_strip.dtype, device=local_strip.device)
    dist.all_gather_into_tensor(out.view(-1), local_strip.contiguous().view(-1))
    return out
