"""Frame-axis shard of ONE long clip over the GPUs of a node (BASELINE configs[3]: 32 frames at 768x768
on 8 GPUs -> 4 frames per rank).

What is per-frame in the I2VGen-XL UNet runs unchanged on the rank's frames: resnets and their 4-D GroupNorms, spatial
self- and cross-attention, up/down-sampling, conv_in / conv_out, every LayerNorm and feed-forward, the spatial Q/K and
feature injections.  Three things cross frames (``pnp_utils.py:170-220``, ``:720-887``, ``:1042-1057``,
``pipeline_i2vgen_xl.py:271-290``): temporal attention (all F frames of K/V per pixel), the temporal conv stacks
(+-1 frame per conv, four chained) and the 5-D GroupNorms in front of both (statistics over all frames).  All three are
point-wise in (h, w), so each *temporal section* runs PIXEL-sharded instead:

    frame shard  [B, F/N, HW,   C]  --exchange-->  pixel shard [B, F, HW/N, C]  -- temporal section --  exchange back

One exchange moves (N-1)/N of the rank's rows once; every kernel of the section (GroupNorm apply, K=3 frame conv,
QKV / out projections, frame attention, GEGLU feed-forward) then sees all F frames of its pixels and needs no halo or
K/V gather.  The only other collective is the 12 x B x groups-byte all-gather of GroupNorm moments.

``exchange``:
  * ``"a2a"``       RCCL all-to-all (default): each rank sends 1/N of its rows to every peer -- on the xGMI mesh every
                    pair of GPUs has its own link, so all seven transfers of a rank proceed concurrently.
  * ``"allgather"`` RCCL all-gather of the whole tensor, then each rank keeps its pixel slab (the form BASELINE.json
                    names for temporal attention).  N x the bytes of ``a2a``; kept for comparison on the 8-GPU node.
With the ``gloo`` backend (CPU tests, or two test processes sharing one GPU) the same collectives
(``all_to_all_single`` / ``all_gather``) run on host copies of the tensors.

``transport``:
  * ``"torch"`` (default) the collectives are ``torch.distributed`` calls (backend ``nccl`` = RCCL on device tensors; ``gloo``
                on host copies for the CPU tests and for two test processes sharing one GPU).
  * ``"rccl"``  the library's own communicator behind the C ABI (``mvoc_allgather_frames`` / ``mvoc_alltoall_frames``,
                ``include/mvoc_hip.h``): torch.distributed only carries the 128-byte unique id.  Device tensors only.
                EXPERIMENTAL: exercised at world size 1 only (the build and test boxes have one GPU) -- per-peer byte
                counts, peer order and the all-to-all signature are unverified across ranks; the default transport is
                ``"torch"``.  Use as a context manager (or call ``close()``) to destroy the communicator.
The pack / unpack copies either side of an exchange are one HIP kernel (``mvoc_permute_rows_f16``) for device tensors and
plain torch for host tensors, so the module stays importable and testable on a CPU-only machine.
"""
import torch
import torch.distributed as dist

__all__ = ["FrameShard"]


class FrameShard:
    def __init__(self, group=None, exchange="a2a", transport="torch", device=None):
        if not dist.is_initialized():
            raise RuntimeError("FrameShard: torch.distributed is not initialised (launch one process per GPU)")
        if exchange not in ("a2a", "allgather"):
            raise ValueError(f"FrameShard: unknown exchange {exchange!r}")
        if transport not in ("torch", "rccl"):
            raise ValueError(f"FrameShard: unknown transport {transport!r}")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.exchange = exchange
        self.bytes_sent = 0  # per-rank payload handed to the collectives since the last reset (accounting only)
        self.transport = transport
        self._comm = None
        self.device = device  # transport "rccl": the GPU the communicator is created on (default: the current device)
        if transport == "rccl":
            self._native_init()

    # ---- the library's own RCCL communicator (C ABI) ----------------------------------------------------
    def _native_init(self):
        import ctypes as C
        from ._ffi import check, lib
        buf = (C.c_char * 128)()
        if self.rank == 0:
            check(lib.mvoc_comm_unique_id(buf), "comm_unique_id")
        box = [bytes(buf)]
        dist.broadcast_object_list(box, src=0, group=self.group)  # the only use of torch.distributed on this transport
        comm = C.c_void_p()
        # ncclCommInitRank binds the communicator to the CURRENT HIP device: make that the engine's device, not whatever the
        # caller's thread happens to have selected
        with torch.cuda.device(self.device if self.device is not None else torch.cuda.current_device()):
            check(lib.mvoc_comm_init(box[0], self.rank, self.world, C.byref(comm)), "comm_init")
        self._comm = comm

    def close(self):
        if self._comm is not None:
            from ._ffi import check, lib
            check(lib.mvoc_comm_destroy(self._comm), "comm_destroy")
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown: the library may already be gone)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _native(self, t):
        if self._comm is None:
            return False
        if not t.is_cuda:
            raise RuntimeError("FrameShard(transport='rccl') exchanges device tensors only")
        return True

    # ---- partition ---------------------------------------------------------------------------------
    def check(self, frames, hw):
        if frames % self.world or hw % self.world:
            raise RuntimeError(f"FrameShard: frames ({frames}) and pixels per frame ({hw}) must both be multiples of the "
                               f"world size {self.world}")

    def frame_range(self, frames):
        n = frames // self.world
        return self.rank * n, (self.rank + 1) * n

    def pixel_range(self, hw):
        n = hw // self.world
        return self.rank * n, (self.rank + 1) * n

    def barrier(self):
        dist.barrier(group=self.group)

    # ---- collectives (device tensors with nccl; staged through the host with gloo) -------------------
    def _staged(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_gather(self, t):
        """[...] -> [world, ...] in rank order"""
        t = t.contiguous()
        self.bytes_sent += t.numel() * t.element_size()
        if self._native(t):
            from ._ffi import check, lib
            res = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            check(lib.mvoc_allgather_frames(self._comm, t.data_ptr(), res.data_ptr(), t.numel() * t.element_size(),
                                            torch.cuda.current_stream().cuda_stream), "allgather_frames")
            return res
        src = t.cpu() if self._staged(t) else t
        out = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(out, src, group=self.group)
        res = torch.stack(out)
        return res.to(t.device) if self._staged(t) else res

    def all_to_all(self, send):
        """send [world, ...]: slice j goes to rank j -> recv [world, ...]: slice i came from rank i"""
        send = send.contiguous()
        assert send.shape[0] == self.world
        if self.exchange == "allgather":
            return self.all_gather(send)[:, self.rank].contiguous()
        self.bytes_sent += send.numel() * send.element_size() * (self.world - 1) // self.world
        if self._native(send):
            from ._ffi import check, lib
            recv = torch.empty_like(send)
            check(lib.mvoc_alltoall_frames(self._comm, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size() // self.world,
                                           torch.cuda.current_stream().cuda_stream), "alltoall_frames")
            return recv
        src = send.cpu() if self._staged(send) else send  # gloo: the same collective on host tensors
        recv = torch.empty_like(src)
        dist.all_to_all_single(recv, src, group=self.group)
        return recv.to(send.device) if self._staged(send) else recv

    # ---- layout exchanges on canonical rows [B * frames * pixels, C] ----------------------------------
    @staticmethod
    def _permute(rows, shape4, perm):
        """rows [prod(shape4), C] viewed as [*shape4, C] -> permuted, contiguous, [*shape4[perm], C]: one HIP kernel for
        device tensors (mvoc_permute_rows_f16), torch for host tensors; the identity costs nothing"""
        c = rows.shape[-1]
        out_shape = tuple(shape4[p] for p in perm) + (c,)
        if tuple(perm) == (0, 1, 2, 3):
            return rows.reshape(out_shape)
        if rows.is_cuda and rows.dtype == torch.float16 and c % 8 == 0:
            from . import ops
            return ops.permute_rows(rows.reshape(-1, c).contiguous(), tuple(shape4), tuple(perm)).view(out_shape)
        return rows.reshape(tuple(shape4) + (c,)).permute(*perm, 4).contiguous()

    def to_pixel_shard(self, x, batch, frames_local, hw):
        """rows (b, f_local, p) of this rank's frames -> rows (b, f, p_local) of ALL frames for this rank's pixel slab"""
        n, c = self.world, x.shape[1]
        hwl = hw // n
        send = self._permute(x, (batch, frames_local, n, hwl), (2, 0, 1, 3))  # [dst, B, Floc, hwl, C]
        recv = self.all_to_all(send)                                            # [src = frame block, B, Floc, hwl, C]
        out = recv if batch == 1 else self._permute(recv.reshape(-1, c), (n, batch, frames_local, hwl), (1, 0, 2, 3))
        return out.reshape(batch * n * frames_local * hwl, c)

    def to_frame_shard(self, y, batch, frames_local, hw):
        """inverse of ``to_pixel_shard``"""
        n, c = self.world, y.shape[1]
        hwl = hw // n
        send = y if batch == 1 else self._permute(y, (batch, n, frames_local, hwl), (1, 0, 2, 3))  # [dst = frame block, B, Floc, hwl, C]
        recv = self.all_to_all(send.reshape(n, batch, frames_local, hwl, c))    # [src = pixel slab, B, Floc, hwl, C]
        return self._permute(recv.reshape(-1, c), (n, batch, frames_local, hwl), (1, 2, 0, 3)).reshape(batch * frames_local * hw, c)

    def gather_frames(self, t, dim):
        """all-gather along the frame dimension ``dim`` (model inputs / outputs, 4 channels: small)"""
        parts = self.all_gather(t)
        return torch.cat(list(parts.unbind(0)), dim=dim)
