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
    """uint8 tensor [row1-row0, cols] aliasing the ASCII output rows of the renderer's most recent frame in
    HBM (no copy).  Overlapping frames alternate between buffer sets: take the alias after the frame."""
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


class StripBatcher:
    """Collects this rank's ASCII band of `batch` consecutive frames in a ring and combines the bands of
    all ranks with ONE all-gather per batch: a collective costs the host tens of microseconds, a frame
    takes less, so gathering frame by frame would leave the GPUs waiting for the host.  Frame k of the
    last gathered batch is frame(k) (every rank holds every full frame).

    `device` tensors live wherever the bands live (HBM with RCCL, host memory with gloo in the tests)."""

    def __init__(self, band_rows, cols, batch, device, before_gather=None, after_gather=None):
        """before_gather / after_gather: called around each all-gather, e.g. to make the gathering stream
        wait for the frames that filled the ring and the next frames wait for the gather (stream-ordered)."""
        self.before_gather, self.after_gather = before_gather, after_gather
        self.world = dist.get_world_size()
        self.batch, self.h, self.cols = int(batch), int(band_rows), int(cols)
        self.ring = torch.empty((self.batch, self.h, self.cols), dtype=torch.uint8, device=device)
        self.out = torch.empty((self.world, self.batch, self.h, self.cols), dtype=torch.uint8, device=device)
        self.filled = 0
        self.gathered = 0      # frames held by `out`

    def push(self, band):
        """Copies one frame's band into the ring (stream-ordered on the current stream); gathers when full."""
        self.ring[self.filled].copy_(band)
        self.filled += 1
        if self.filled == self.batch:
            self.flush()

    def slot_ptr(self):
        """Device address of the ring slot the next frame's band belongs in (for ar_set_ascii_target)."""
        return self.ring.data_ptr() + self.filled * self.h * self.cols

    def advance(self):
        """The renderer has been told to write the band into slot_ptr() itself (no copy); gathers when full."""
        self.filled += 1
        if self.filled == self.batch:
            self.flush()

    def flush(self):
        if self.filled == 0:
            return
        if self.before_gather:
            self.before_gather()
        dist.all_gather_into_tensor(self.out.view(-1), self.ring.view(-1))
        if self.after_gather:
            self.after_gather()
        self.gathered, self.filled = self.filled, 0

    def frame(self, k):
        """Full grid [world * band_rows, cols] of frame k of the last gathered batch."""
        if not 0 <= k < self.gathered:
            raise IndexError("frame %d of a batch of %d" % (k, self.gathered))
        return self.out[:, k].reshape(self.world * self.h, self.cols)
