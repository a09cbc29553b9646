"""RCCL through the C ABI (include/city2ba_hip.h: c2b_comm_*): the collectives of the sharded path as a Rust host
would use them -- no torch.distributed in the data path.  torch is only the messenger that carries the 128-byte
communicator id from rank 0 to the other ranks when a process group already exists (any backend); a host without
torch passes the id through a file, a pipe or MPI instead."""
import ctypes as C

from . import _lib as L


def backend():
    return L.lib().c2b_comm_backend().decode("utf-8", "replace")


def unique_id():
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    L.check(L.lib().c2b_comm_unique_id(buf))
    return bytes(buf.raw)


class Comm:
    """one rank's membership of a communicator; collectives run on the CURRENT torch stream of `device`"""

    def __init__(self, comm_id, rank, world, device):
        self._h = C.c_void_p()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        buf = C.create_string_buffer(bytes(comm_id), L.COMM_ID_BYTES)
        L.check(L.lib().c2b_comm_init_rank(buf, self.rank, self.world, self.device, C.byref(self._h)))

    @classmethod
    def from_process_group(cls, device, group=None):
        """every rank of an initialised torch.distributed group joins one communicator: rank 0 makes the id, the
        group's object broadcast carries it.  No rank can be left waiting in a collective the others never enter:
        (1) every rank probes the backend (RCCL loadable?) and the answers are gathered BEFORE anyone makes an id or calls
        init_rank -- one rank without RCCL and every rank raises together; (2) rank 0 broadcasts either the id or the
        text of its failure, so a failing c2b_comm_unique_id reaches every rank as an exception instead of a hang;
        (3) the outcome of c2b_comm_init_rank is gathered, so a rank whose init failed takes every rank down with it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def everyone(ok, why):
            """gather (ok, why) from every rank; raise on all of them if any failed"""
            if world == 1:
                answers = [(ok, why)]
            else:
                answers = [None] * world
                dist.all_gather_object(answers, (bool(ok), str(why)), group=group)
            bad = ["rank %d: %s" % (r, w) for r, (o, w) in enumerate(answers) if not o]
            if bad:
                raise L.City2baError(L.ERR_RCCL, "communicator unavailable -- " + "; ".join(bad))

        name = backend()
        everyone(name.startswith("RCCL"), "c2b_comm_backend() = %r" % name)
        box = [None]
        if rank == 0:
            try:
                box = [("id", unique_id())]
            except Exception as exc:                                  # noqa: BLE001 -- travels to every rank
                box = [("error", "%s: %s" % (type(exc).__name__, exc))]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        kind, payload = box[0]
        if kind != "id":
            raise L.City2baError(L.ERR_RCCL, "rank 0 could not make a communicator id -- " + payload)
        self, why = None, ""
        try:
            self = cls(payload, rank, world, device)
        except Exception as exc:                                      # noqa: BLE001
            why = "%s: %s" % (type(exc).__name__, exc)
        try:
            everyone(self is not None, why)
        except Exception:
            if self is not None:
                self.destroy()
            raise
        return self

    def info(self):
        """(rank, world, device) as the communicator itself reports them (c2b_comm_info)"""
        r, w, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        L.check(L.lib().c2b_comm_info(self._h, C.byref(r), C.byref(w), C.byref(d)))
        return r.value, w.value, d.value

    @property
    def handle(self):
        return self._h

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce_sum_(self, t):
        import torch
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.device.index == self.device
        L.check(L.lib().c2b_comm_all_reduce_sum_f64(self._h, C.c_void_p(t.data_ptr()), t.numel(), self._stream()))
        return t

    def all_gather(self, t):
        import torch
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.device.index == self.device
        out = torch.empty((self.world,) + tuple(t.shape), dtype=torch.float64, device=t.device)
        L.check(L.lib().c2b_comm_all_gather_f64(self._h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(out.data_ptr()),
                                                self._stream()))
        return out

    def stats_sharded(self, camblk, cam_base, n_cam_global, pts4, ws, out=None, centers=None):
        """c2b_stats_sharded: the statistics record over sharded cameras, identical bits on every rank"""
        import torch
        out = out if out is not None else torch.empty(L.STATS_DOUBLES, dtype=torch.float64, device=camblk.device)
        if centers is not None and (tuple(centers.shape) != (camblk.shape[0], 4) or centers.dtype != torch.float64):
            raise ValueError("centers must be the [n_cam][4] f64 table cameras_prepare_* filled")
        L.check(L.lib().c2b_stats_sharded(self._h, C.c_void_p(camblk.data_ptr()),
                                          C.c_void_p(centers.data_ptr()) if centers is not None else None, camblk.shape[0], int(cam_base),
                                          int(n_cam_global), C.c_void_p(pts4.data_ptr()), pts4.shape[0],
                                          C.c_void_p(ws.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def destroy(self):
        h, self._h = self._h, C.c_void_p()
        if h:
            L.lib().c2b_comm_destroy(h)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
