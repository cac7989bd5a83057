"""Worker of the launcher-contract tests: started N times as a PLAIN process — by `mpiexec -np N`, or by the test with one launcher's environment
variables set — it takes rank and size from the environment the way the drop-in executable does (minimd_amd.api.launch_env: ref/ljs.cpp:63-68 asks
MPI), meets the other ranks on the built-in TCP mesh (csrc/launch.cpp) and pushes real messages through it along Comm::setup's swap pattern.
No GPU, no torch.   usage: mesh_worker.py <outdir>"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
from minimd_amd import api


def main(outdir):
    le = api.launch_env()
    rank, world = le["rank"], le["nranks"]
    addr, port = api.launch_rendezvous()
    mesh = api.Mesh(rank, world, addr, port)
    errors = []
    # Comm::setup geometry of every rank (host-only handle), gathered over the mesh
    dims = (8, 6, 10)
    prd = api.create_box(*dims, 0.8442)
    h = minimd_amd.Handle(device=-2)
    h.set_box(prd)
    h.comm_setup(2.8, rank, world)
    info = h.comm_info()
    swaps = [h.swap_info(s) for s in range(info["nswap"])]
    mine = np.array([[sw["sendproc"], sw["recvproc"]] for sw in swaps], dtype=np.int32).tobytes()
    allsw = [np.frombuffer(b, dtype=np.int32).reshape(-1, 2) for b in mesh.allgather(mine)]
    for s, sw in enumerate(swaps):
        if allsw[sw["sendproc"]][s][1] != rank:
            errors.append("rank %d swap %d: sendproc/recvproc mismatch" % (rank, s))
    # real bytes along the swap pattern: count handshake, then a payload of rank-dependent length (0.3 to 3 MB: larger than the socket buffers)
    for s, sw in enumerate(swaps):
        n = 40000 + 170000 * ((rank + s) % 3) + 37 * rank + s
        payload = (np.arange(n, dtype=np.int64) * (rank + 1) + s).tobytes()
        cnt = np.frombuffer(mesh.sendrecv(np.int32(n).tobytes(), sw["sendproc"], 4, sw["recvproc"]), dtype=np.int32)[0]
        got = np.frombuffer(mesh.sendrecv(payload, sw["sendproc"], int(cnt) * 8, sw["recvproc"]), dtype=np.int64)
        src = sw["recvproc"]
        nexp = 40000 + 170000 * ((src + s) % 3) + 37 * src + s
        exp = np.arange(nexp, dtype=np.int64) * (src + 1) + s
        if len(got) != len(exp) or not np.array_equal(got, exp):
            errors.append("rank %d swap %d: payload mismatch (%d / %d)" % (rank, s, len(got), len(exp)))
    # one-sided messages (the direct halo pairs "send k" with "receive k" and either may be empty)
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    if world > 1:
        if rank % 2 == 0:
            r = mesh.sendrecv(b"x" * (1000 + rank), nxt, 0, rank)
            if r != b"":
                errors.append("rank %d: empty receive returned bytes" % rank)
        if prv % 2 == 0:
            r = mesh.sendrecv(b"", rank, 5000, prv)
            if r != b"x" * (1000 + prv):
                errors.append("rank %d: one-sided message from %d wrong (%d bytes)" % (rank, prv, len(r)))
    # MPI_Allreduce(SUM): same bits on every rank, values that do not commute in floating point
    v = np.array([0.1 * (rank + 1), 1e16 if rank == 0 else 1.0, -1e16 if rank == world - 1 and world > 1 else 0.5])
    mesh.allreduce(v)
    allv = [np.frombuffer(b, dtype=np.float64) for b in mesh.allgather(v.tobytes())]
    for r_ in range(world):
        if not np.array_equal(allv[r_], allv[0]):
            errors.append("allreduce: rank %d and rank 0 hold different bits" % r_)
    if abs(v[0] - sum(0.1 * (r_ + 1) for r_ in range(world))) > 1e-12:
        errors.append("allreduce: wrong sum")
    st = mesh.info()
    json.dump({"rank": rank, "world": world, "launcher": le["launcher"], "local_rank": le["local_rank"], "errors": errors, "port": port,
               "procgrid": info["procgrid"].tolist(), "messages": st["messages"], "bytes_sent": st["bytes_sent"]},
              open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    mesh.close()


if __name__ == "__main__":
    main(sys.argv[1])
