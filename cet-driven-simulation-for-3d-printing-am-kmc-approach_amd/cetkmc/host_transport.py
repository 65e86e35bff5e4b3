"""Host-relay transport for ``Engine(..., host_comm=...)``: the engine's three collectives carried by
``torch.distributed`` (any backend with CPU tensors, normally gloo).  A bring-up / test path -- it synchronises the
GPU stream around every exchange -- that runs the same per-rank device code as the RCCL communicator and lets
several ranks share one GPU.  The process group must be initialised by the caller."""
import ctypes as C


def torch_callbacks():
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()

    def _view(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)))

    def allgather(user, send, recv, nbytes):
        try:
            mine = _view(send, nbytes).clone()
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine)
            _view(recv, nbytes * world).copy_(torch.cat(parts))
            return 0
        except Exception as exc:            # never raise through the C frame
            print("host_transport.allgather:", exc, flush=True)
            return 1

    def exchange(user, lo, hi, send_lo, recv_lo, send_hi, recv_hi, nbytes):
        try:
            ops, keep = [], []
            for peer, sp, rp in ((lo, send_lo, recv_lo), (hi, send_hi, recv_hi)):
                if peer < 0:
                    continue
                s, r = _view(sp, nbytes).clone(), torch.empty(nbytes, dtype=torch.uint8)
                keep.append((r, rp))
                ops += [dist.P2POp(dist.isend, s, peer), dist.P2POp(dist.irecv, r, peer)]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            for r, rp in keep:
                _view(rp, nbytes).copy_(r)
            return 0
        except Exception as exc:
            print("host_transport.exchange:", exc, flush=True)
            return 1

    return allgather, exchange
