"""Host-staged transport for mmd_comm_set_host_transport(): carries the library's halo / migration messages
over a torch.distributed process group (gloo on CPU). Used by the multi-process tests (two ranks sharing one
GPU, or no GPU at all for the pure message-pattern test) and as a fallback when RCCL cannot be used; the
production data path is RCCL inside the library (mmd_comm_init_rccl)."""
import numpy as np
import torch
import torch.distributed as dist


class GlooTransport:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)

    def sendrecv(self, data: bytes, dest: int, nrecv: int, src: int) -> bytes:
        """MPI_Sendrecv semantics: send `data` to dest while receiving exactly nrecv bytes from src."""
        if dest == self.rank and src == self.rank:
            return data[:nrecv]
        reqs = []
        recv_t = torch.empty(max(nrecv, 1), dtype=torch.uint8)
        if nrecv:
            reqs.append(dist.irecv(recv_t[:nrecv], src=src, group=self.group))
        if len(data):
            send_t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
            reqs.append(dist.isend(send_t, dst=dest, group=self.group))
        for r in reqs:
            r.wait()
        return recv_t[:nrecv].numpy().tobytes() if nrecv else b""

    def allreduce(self, arr: np.ndarray) -> None:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        arr[:] = t.numpy()

    def attach(self, handle):
        handle.set_host_transport(self.sendrecv, self.allreduce)
