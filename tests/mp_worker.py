"""Worker for the multi-process tests, launched by torch.distributed.run (gloo, 127.0.0.1).
  mode geometry : CPU only — every rank builds its Comm::setup geometry with a host-only handle, the ranks
                  cross-check their swap partners / slabs and push real buffers through GlooTransport along
                  exactly that swap pattern (sendrecv + allreduce semantics of the library's transport).
  mode sim      : GPU — every rank runs the product (ranks share GPU 0 when only one is visible) with the
                  gloo host transport; rank 0 writes the thermo rows + per-rank counts to <out>.
  mode simrccl  : the same over the library's RCCL communicator (one GPU per rank; needs as many GPUs as ranks).
"""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
from minimd_amd import api
from minimd_amd.transport import GlooTransport


def geometry(out):
    rank, world = dist.get_rank(), dist.get_world_size()
    dims = (8, 6, 10)
    prd = api.create_box(*dims, 0.8442)
    h = minimd_amd.Handle(device=-2)
    h.set_box(prd)
    h.comm_setup(2.8, rank, world)
    info = h.comm_info()
    swaps = [h.swap_info(s) for s in range(info["nswap"])]
    _, lo, hi = h.get_box()
    mine = {"rank": rank, "procgrid": info["procgrid"].tolist(), "myloc": info["myloc"].tolist(), "lo": lo.tolist(), "hi": hi.tolist(),
            "swaps": swaps}
    allinfo = [None] * world
    dist.all_gather_object(allinfo, mine)
    errors = []
    # partner symmetry: if I send swap s to p, then p receives swap s from me
    for s, sw in enumerate(swaps):
        peer = allinfo[sw["sendproc"]]
        if peer["swaps"][s]["recvproc"] != rank:
            errors.append("swap %d: sendproc/recvproc mismatch" % s)
    vol = sum(np.prod(np.array(a["hi"]) - np.array(a["lo"])) for a in allinfo)
    if abs(vol - np.prod(prd)) > 1e-9 * np.prod(prd):
        errors.append("sub-boxes do not tile the box")
    # real bytes along the swap pattern
    tr = GlooTransport()
    for s, sw in enumerate(swaps):
        n = 1000 + 37 * rank + s
        payload = (np.arange(n, dtype=np.int64) * (rank + 1) + s).tobytes()
        cnt = np.frombuffer(tr.sendrecv(np.int32(n).tobytes(), sw["sendproc"], 4, sw["recvproc"]), dtype=np.int32)[0]
        got = np.frombuffer(tr.sendrecv(payload, sw["sendproc"], int(cnt) * 8, sw["recvproc"]), dtype=np.int64)
        src = sw["recvproc"]
        exp = np.arange(1000 + 37 * src + s, dtype=np.int64) * (src + 1) + s
        if len(got) != len(exp) or not np.array_equal(got, exp):
            errors.append("swap %d: payload mismatch" % s)
    v = np.array([rank + 1.0, 10.0 * (rank + 1), 0.5])
    tr.allreduce(v)
    exp = np.array([sum(r + 1.0 for r in range(world)), sum(10.0 * (r + 1) for r in range(world)), 0.5 * world])
    if not np.allclose(v, exp):
        errors.append("allreduce mismatch")
    allerr = [None] * world
    dist.all_gather_object(allerr, errors)
    if rank == 0:
        json.dump({"errors": sum(allerr, []), "procgrid": mine["procgrid"], "world": world}, open(out, "w"))


def sim(out, args, precision, rccl=False):
    rank, world = dist.get_rank(), dist.get_world_size()
    if rccl:
        # production transport: the library's own RCCL communicator, one GPU per rank (LOCAL_RANK picks the device)
        import ctypes
        L = api.load_library(precision)
        obj = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            assert L.mmd_comm_unique_id(buf) == 0, L.mmd_last_error()
            obj = [buf.raw]
        dist.broadcast_object_list(obj, src=0)
        L.mmd_sim_set_unique_id(obj[0])
    else:
        tr = GlooTransport()
        api.sim_set_host_transport(tr.sendrecv, tr.allreduce, precision)
    s = minimd_amd.Sim(args, precision=precision)
    for kv in filter(None, os.environ.get("MMD_TEST_OPTIONS", "").split(",")):      # e.g. borders_est=60: undersized ghost arrays
        k, v = kv.split("=")
        s.handle.set_option(k, int(v))
    s.initial()
    s.run()
    nl, ng, _ = s.handle.counts()
    counts = [None] * world
    dist.all_gather_object(counts, (nl, ng, s.handle.neighbor_info()["total"]))
    stats = [None] * world
    st = s.handle.run_stats()                                    # of Integrate::run (the last mmd_integrate_run of Sim.run)
    for c in ("exchange_fast", "exchange_overflows", "borders_fast", "borders_general", "borders_direct"):
        st[c] = s.handle.counter(c)
    rows = s.rows()
    steady = int(os.environ.get("MMD_TEST_STEADY", "0"))
    if steady > 0:
        # a further slice behind the run (its plans exist: fixed-size messages sized from the previous re-neighboring): the steady-state cost per re-neighboring
        s.run_steps(steady)
        st2 = s.handle.run_stats()
        st["steady"] = {"steps": steady, "host_syncs": st2["host_syncs"], "bytes_sent": st2["bytes_sent"],
                        "exchange_overflows": s.handle.counter("exchange_overflows"), "borders_general": s.handle.counter("borders_general")}
    dist.all_gather_object(stats, st)
    if rank == 0:
        json.dump({"rows": rows, "counts": counts, "natoms": s.natoms(), "stats": stats}, open(out, "w"))
    s.close()


def exchange_all(out, args, precision):
    """GPU, ranks share the visible GPU(s), gloo host transport: after Sim.initial() every rank pushes some of its atoms two
    sub-domains away (every 5th up, every 5th+1 down, in the last dimension of the grid), then runs Comm::exchange with the
    safe-exchange option (Comm::exchange_all, ref/comm.cpp:599-689). Rank 0 writes every rank's owned positions, in order."""
    rank, world = dist.get_rank(), dist.get_world_size()
    tr = GlooTransport()
    api.sim_set_host_transport(tr.sendrecv, tr.allreduce, precision)
    s = minimd_amd.Sim(args, precision=precision)
    s.initial()
    h = s.handle
    d = h.download()
    nl = d["nlocal"]
    prd, lo, hi = h.get_box()
    info = h.comm_info()
    dim = int(np.argmax(info["procgrid"]))
    x = d["x"][:nl].copy()
    w = hi[dim] - lo[dim]
    idx = np.arange(nl)
    x[idx % 5 == 0, dim] += 2 * w
    x[idx % 5 == 1, dim] -= 2 * w
    h.upload(x, d["v"], d["type"][:nl], d["tag"])
    h.set_option("safe_exchange", int(os.environ.get("MMD_TEST_SAFE", "1")))
    h.exchange()
    e = h.download()
    mine = {"x": e["x"][:e["nlocal"]].tolist(), "v": e["v"].tolist(), "dim": dim, "need": info["need"].tolist(), "procgrid": info["procgrid"].tolist()}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank == 0:
        json.dump(allr, open(out, "w"))
    s.close()


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    dist.init_process_group(backend="gloo")
    try:
        if mode == "geometry":
            geometry(out)
        elif mode == "exchall":
            exchange_all(out, sys.argv[4:], sys.argv[3])
        else:
            sim(out, sys.argv[4:], sys.argv[3], rccl=(mode == "simrccl"))
        dist.barrier()
    finally:
        dist.destroy_process_group()
