"""Screen-strip partition for multi-GPU runs: one process per GPU, each renders a contiguous band of
output rows from its own replica of the mesh; the ASCII bands are combined with one all-gather per
frame (torch.distributed: RCCL on GPUs, gloo in the CPU tests).  The reference is single-GPU; this
is the build's own scaling layer on top of the C ABI's ar_set_strip."""
import torch
import torch.distributed as dist


def strip_bounds(rows, world, rank):
    """Rows [row0, row1) of rank `rank` when `rows` rows are cut into `world` bands: heights differ by at most one row
    (the first rows % world bands are the taller ones), so any number of rows goes with any number of ranks - the
    arithmetic of ar_strip_rows.  The all-gather moves slots of the tallest band's height (band_height)."""
    if world <= 0 or not 0 <= rank < world or rows < 0:
        raise ValueError("bad rank %d of %d" % (rank, world))
    base, rem = divmod(rows, world)
    row0 = rank * base + min(rank, rem)
    return row0, row0 + base + (1 if rank < rem else 0)


def band_height(rows, world):
    """Rows of the tallest band: the slot height every rank contributes to the all-gather."""
    return (rows + world - 1) // world


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

    There are `rings` rings, used in rotation (two by default): while one is being gathered the frames of the
    next batch already fill the other, so neither the renderer nor the interconnect waits for the other.

    `device` tensors live wherever the bands live (HBM with RCCL, host memory with gloo in the tests)."""

    def __init__(self, band_rows, cols, batch, device, before_gather=None, after_gather=None, rings=2, rows=None):
        """band_rows: rows of a slot - this rank's band when the frame divides evenly, band_height(rows, world) otherwise,
        with `rows` (the frame's) given so that frame() can cut the padding out again.  before_gather: called right before each all-gather, to make the gathering stream wait for the frames
        that filled the ring.  after_gather: makes the frames enqueued from then on wait for the gathers
        enqueued so far (stream-ordered, no host wait); with one ring it is called right after each gather, with
        more it is called before the next gather is enqueued - the frames that follow reuse the ring of an
        earlier gather, not of the one about to start."""
        self.before_gather, self.after_gather = before_gather, after_gather
        self.world = dist.get_world_size()
        self.batch, self.h, self.cols, self.rings = int(batch), int(band_rows), int(cols), max(1, int(rings))
        self.rows = self.world * self.h if rows is None else int(rows)
        if band_height(self.rows, self.world) != self.h:
            raise ValueError("slots of %d rows do not fit %d rows over %d ranks" % (self.h, self.rows, self.world))
        self.ring = torch.empty((self.rings, self.batch, self.h, self.cols), dtype=torch.uint8, device=device)
        self.out = torch.empty((self.rings, self.world, self.batch, self.h, self.cols), dtype=torch.uint8, device=device)
        self.cur = 0           # ring being filled
        self.last = 0          # ring of the last gathered batch
        self.filled = 0
        self.gathered = 0      # frames held by out[last]

    def push(self, band):
        """Copies one frame's band (its own rows; shorter than the slot for the lower ranks of an uneven cut) into the
        ring (stream-ordered on the current stream); gathers when full."""
        self.ring[self.cur, self.filled, :band.shape[0]].copy_(band)
        self.filled += 1
        if self.filled == self.batch:
            self.flush()

    def slot_ptr(self):
        """Device address of the ring slot the next frame's band belongs in (for ar_set_ascii_target)."""
        return self.ring.data_ptr() + (self.cur * self.batch + self.filled) * self.h * self.cols

    def advance(self):
        """The renderer has been told to write the band into slot_ptr() itself (no copy); gathers when full."""
        self.filled += 1
        if self.filled == self.batch:
            self.flush()

    def flush(self):
        if self.filled == 0:
            return
        if self.after_gather and self.rings > 1:
            self.after_gather()
        if self.before_gather:
            self.before_gather()
        dist.all_gather_into_tensor(self.out[self.cur].view(-1), self.ring[self.cur].view(-1))
        if self.after_gather and self.rings == 1:
            self.after_gather()
        self.last, self.gathered, self.filled = self.cur, self.filled, 0
        self.cur = (self.cur + 1) % self.rings

    def frame(self, k):
        """Full grid [rows, cols] of frame k of the last gathered batch (slot padding of an uneven cut removed)."""
        if not 0 <= k < self.gathered:
            raise IndexError("frame %d of a batch of %d" % (k, self.gathered))
        if self.rows == self.world * self.h:
            return self.out[self.last][:, k].reshape(self.world * self.h, self.cols)
        bands = []
        for r in range(self.world):
            row0, row1 = strip_bounds(self.rows, self.world, r)
            bands.append(self.out[self.last][r, k, :row1 - row0])
        return torch.cat(bands, dim=0)
