"""Row tiling of the framebuffer across the GPUs of one node + the single gather of the
composed image (BASELINE north_star). No reference counterpart: the reference is
single-GPU. Pixels are independent, so there is no halo and no other collective.

Rows are cut into blocks of `block_rows`; block b belongs to rank b % nranks (cyclic), which
balances the cheap sky rows at the top of a frame against the expensive geometry rows at the
bottom. A rank stores only its own rows, packed; include/szg/abi.h `szg_rowtile` gives the
local -> global row map the kernels use for the camera rays (camera.comp:324 uses global pixel
coordinates and the full draw extent).

One process per GPU. The product's collectives are the C-ABI's (`Comm` below = szg_rowtile_comm of
include/szg/abi.h: RCCL over xGMI, N-1 point-to-point streams into rank 0 for the tile gather and an in-place
all-gather for the sky-view LUT slices), followed by one HBM-bound row scatter (`szg_compose_rowtiles`) on
rank 0. The functions gather_tiles / allgather_lut do the same through torch.distributed and exist for the gloo
rehearsals (CPU tests; several ranks sharing one GPU, where RCCL refuses to run).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import abi
from ._lib import check, lib

DEFAULT_BLOCK_ROWS = 8


def local_rows(height, rank, nranks, block_rows=DEFAULT_BLOCK_ROWS):
    return int(lib().szg_rowtile_local_rows(int(height), int(block_rows), int(rank), int(nranks)))


def make_tile(height, rank, nranks, block_rows=DEFAULT_BLOCK_ROWS):
    """abi.RowTile for `rank`, or None when there is a single rank (whole frame)."""
    if nranks <= 1:
        return None
    return abi.RowTile(int(block_rows), int(rank), int(nranks), local_rows(height, rank, nranks, block_rows))


def stride_rows(height, nranks, block_rows=DEFAULT_BLOCK_ROWS):
    """Rows of the largest tile: every rank's colour buffer is padded to this so that the gather
    moves equal-sized tensors."""
    return max(local_rows(height, r, nranks, block_rows) for r in range(max(nranks, 1)))


def global_rows(height, rank, nranks, block_rows=DEFAULT_BLOCK_ROWS):
    """Global row index of each local row of `rank` (host mirror of szg_rowtile's map)."""
    if nranks <= 1:
        return np.arange(height, dtype=np.int64)
    nblocks = (height + block_rows - 1) // block_rows
    out = [np.arange(b * block_rows, min(height, (b + 1) * block_rows), dtype=np.int64) for b in range(rank, nblocks, nranks)]
    return np.concatenate(out) if out else np.zeros((0,), np.int64)


class _EventWork:
    """Work handle of a collective enqueued on a side stream: wait() makes the CURRENT stream wait for it (the host
    never blocks), like torch.distributed's Work.wait() for RCCL."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


class Comm:
    """szg_rowtile_comm (abi.h "Multi-GPU collectives"): this rank's RCCL communicators behind the C-ABI. Creating it is
    a collective call. The opaque id travels from rank 0 to the others through torch.distributed's object broadcast on
    `bootstrap_group` (any backend: the control plane may be gloo); a C++ caller would use its own launcher for that.
    The two collectives run on two side streams of their own, ordered against the compute stream by events."""

    def __init__(self, rank, nranks, device_index=0, bootstrap_group=None):
        self.rank, self.nranks = int(rank), int(nranks)
        blob = C.create_string_buffer(abi.SZG_ROWTILE_COMM_ID_BYTES)
        failure = None
        if self.rank == 0:
            try:
                check(lib().szg_rowtile_comm_unique_id(blob))
            except Exception as e:  # RCCL missing: tell the other ranks instead of leaving them in the broadcast
                failure = e
        if self.nranks > 1:
            box = [b"" if failure is not None else bytes(blob.raw)]
            dist.broadcast_object_list(box, src=0, group=bootstrap_group)
            if len(box[0]) != abi.SZG_ROWTILE_COMM_ID_BYTES:
                raise RuntimeError(f"rank 0 could not create the communicator id: {failure}")
            blob = C.create_string_buffer(box[0], abi.SZG_ROWTILE_COMM_ID_BYTES)
        elif failure is not None:
            raise failure
        handle = C.c_void_p()
        check(lib().szg_rowtile_comm_create(C.byref(handle), self.rank, self.nranks, blob, int(device_index)))
        self._h = handle
        self.device = torch.device("cuda", int(device_index))
        self.tile_stream = torch.cuda.Stream(self.device)
        self.lut_stream = torch.cuda.Stream(self.device)

    def size(self):
        """The rank count RCCL itself reports for the communicator."""
        return int(lib().szg_rowtile_comm_size(self._h))

    def destroy(self):
        if self._h is not None:
            lib().szg_rowtile_comm_destroy(self._h)
            self._h = None

    def gather_tiles(self, local_color, gathered=None, dst=0):
        """THE collective of the frame (szg_rowtile_gather). `local_color`: [stride_rows, W, 4] int16 on every rank;
        `gathered`: [nranks, stride_rows, W, 4] on `dst` (allocated if None). Returns (gathered or None, work): wait()
        before reading `gathered` or overwriting `local_color`."""
        assert local_color.is_cuda and local_color.is_contiguous()
        if self.rank == dst and gathered is None:
            gathered = torch.empty((self.nranks,) + tuple(local_color.shape), dtype=local_color.dtype, device=local_color.device)
        self.tile_stream.wait_stream(torch.cuda.current_stream())  # the tile was rendered on the compute stream
        nbytes = local_color.numel() * local_color.element_size()
        check(lib().szg_rowtile_gather(self._h, C.c_void_p(self.tile_stream.cuda_stream), C.c_void_p(local_color.data_ptr()), nbytes,
                                       C.c_void_p(gathered.data_ptr()) if self.rank == dst else None, int(dst)))
        done = torch.cuda.Event()
        done.record(self.tile_stream)
        return (gathered if self.rank == dst else None), _EventWork(done)

    def allgather_skyview_lut(self, sky):
        """The optional second collective (szg_skyview_allgather_lut_rows): every rank has recorded its row slice
        (sky.lutRowSlice) on the current stream; exchange the slices in place in the pipeline's own LUT memory."""
        self.lut_stream.wait_stream(torch.cuda.current_stream())
        check(lib().szg_skyview_allgather_lut_rows(sky._h, self._h, C.c_void_p(self.lut_stream.cuda_stream)))
        done = torch.cuda.Event()
        done.record(self.lut_stream)
        return _EventWork(done)


class _Done:
    """Completed work handle (the host-staged gloo rehearsal path is synchronous)."""

    def wait(self):
        return True


def _host_staged(tensor, group):
    """gloo moves CUDA tensors only for some collectives: when N ranks are REHEARSED over gloo (CPU tests, or several
    ranks sharing one GPU where RCCL refuses to run), the collectives below stage through host memory. RCCL never does."""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def gather_tiles(local_color, rank, nranks, gathered=None, dst=0, group=None, async_op=False):
    """The one collective of the path. `local_color`: [stride_rows, W, 4] int16 (RGBA16 UNORM bits)
    on every rank; `gathered`: [nranks, stride_rows, W, 4] on `dst` (allocated if None). Returns
    `gathered` on dst, None elsewhere; with async_op=True returns (gathered_or_None, work) and the caller
    must work.wait() before touching `gathered` or overwriting `local_color`."""
    # RCCL and gloo have no 16-bit integer type: move the tiles as bytes
    send = local_color.contiguous().view(torch.uint8)
    if rank == dst and gathered is None:
        gathered = torch.empty((nranks,) + tuple(local_color.shape), dtype=local_color.dtype, device=local_color.device)
    if _host_staged(send, group):
        host = send.cpu()
        if rank == dst:
            parts = [torch.empty_like(host) for _ in range(nranks)]
            dist.gather(host, parts, dst=dst, group=group)
            gathered.view(torch.uint8).copy_(torch.stack(parts))
        else:
            dist.gather(host, None, dst=dst, group=group)
        out = gathered if rank == dst else None
        return (out, _Done()) if async_op else out
    if rank == dst:
        work = dist.gather(send, list(gathered.view(torch.uint8).unbind(0)), dst=dst, group=group, async_op=async_op)
        return (gathered, work) if async_op else gathered
    work = dist.gather(send, None, dst=dst, group=group, async_op=async_op)
    return (None, work) if async_op else None


def compose(gathered, height, nranks, block_rows=DEFAULT_BLOCK_ROWS, out=None, stream=None):
    """Scatter the gathered tiles' rows into the full-frame RGBA16 image on the GPU."""
    if not gathered.is_cuda:
        raise RuntimeError("compose() runs the HIP row-scatter kernel and needs CUDA tensors; there is no CPU fallback")
    n, srows, width, _ = gathered.shape
    assert n == nranks and gathered.is_contiguous()
    if out is None:
        out = torch.empty((height, width, 4), dtype=gathered.dtype, device=gathered.device)
    im = abi.Image(out.data_ptr(), width, height, width * 8, abi.SZG_FORMAT_RGBA16_UNORM)
    handle = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    check(lib().szg_compose_rowtiles(C.c_void_p(handle), C.c_void_p(gathered.data_ptr()), srows * width * 8, nranks, block_rows,
                                     C.byref(im), width, height))
    return out


def lut_rows(height, rank, nranks):
    """[begin, end) rows of a LUT of `height` rows that `rank` computes (contiguous, equal: the all-gather
    needs equal slices, so height must divide by nranks)."""
    if height % max(nranks, 1) != 0:
        raise ValueError(f"LUT height {height} is not divisible by {nranks} ranks")
    n = height // max(nranks, 1)
    return rank * n, (rank + 1) * n


def allgather_lut(lut, rank, nranks, group=None, async_op=False, force=False):
    """Second (optional) collective: every rank has filled rows lut_rows(...) of `lut` ([H, W, 4] float32,
    the same buffer on every rank); exchange the slices in place so that every rank holds the whole LUT.
    Returns the work handle when async_op=True (wait() before reading the LUT)."""
    if nranks <= 1 and not force:
        return None
    b, e = lut_rows(lut.shape[0], rank, nranks)
    if _host_staged(lut, group):
        whole = torch.empty(lut.shape, dtype=lut.dtype)
        dist.all_gather_into_tensor(whole.view(-1), lut[b:e].reshape(-1).cpu(), group=group)
        lut.copy_(whole)
        return _Done() if async_op else None
    return dist.all_gather_into_tensor(lut.view(-1), lut[b:e].reshape(-1).clone(), group=group, async_op=async_op)
