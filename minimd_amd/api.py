"""ctypes mirror of include/mmd.h. Class/method names follow the reference's classes
(Atom, Neighbor, Force, Comm, Integrate, Thermo of Mantevo/miniMD ref/) so tests read like the reference."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
DATA_DIR = os.path.join(REPO, "data")
_LIBS = {}


class MMDError(RuntimeError):
    pass


def lib_path(precision="dp"):
    # MMD_LIB_DIR: tuning builds of the same library kept next to the product one (tools/build_variant.sh)
    return os.path.join(os.environ.get("MMD_LIB_DIR") or os.path.join(PKG, "lib"), "libmmd_hip_%s.so" % precision)


def build(verbose=False):
    """compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)"""
    r = subprocess.run(["make", "-j8", "-C", os.path.join(PKG, "csrc"), "all"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise MMDError("build failed:\n" + (r.stdout or "") + (r.stderr or ""))


class _Input(C.Structure):
    pass


def _input_struct(creal):
    class MMDInput(C.Structure):
        _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("t_request", creal), ("rho", creal),
                    ("units", C.c_int), ("forcetype", C.c_int), ("epsilon", creal), ("sigma", creal),
                    ("datafile", C.c_char * 1000), ("has_datafile", C.c_int), ("ntimes", C.c_int), ("dt", creal),
                    ("neigh_every", C.c_int), ("force_cut", creal), ("neigh_cut", creal), ("thermo_nstat", C.c_int)]
    return MMDInput


def load_library(precision="dp"):
    """dlopen the product library; fails loudly when it has not been built (no fallback of any kind)."""
    if precision in _LIBS:
        return _LIBS[precision]
    path = lib_path(precision)
    if not os.path.exists(path):
        raise MMDError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C minimd_amd/csrc)" % path)
    L = C.CDLL(path)
    creal = C.c_double if precision == "dp" else C.c_float
    P, I, D = C.c_void_p, C.c_int, C.c_double
    rp, ip, dp = C.POINTER(creal), C.POINTER(I), C.POINTER(D)
    L.mmd_last_error.restype = C.c_char_p
    L.mmd_variant_string.restype = C.c_char_p
    sig = {
        "mmd_create": [I, C.POINTER(P)], "mmd_destroy": [P], "mmd_float_size": [],
        "mmd_device_info": [P, C.c_char_p, I, ip, dp],
        "mmd_atom_set_box": [P, rp, rp, rp], "mmd_atom_get_box": [P, rp, rp, rp], "mmd_atom_set_mass": [P, creal],
        "mmd_atom_upload": [P, rp, rp, ip, ip, I, I], "mmd_atom_download": [P, rp, rp, rp, ip, ip],
        "mmd_atom_upload_f": [P, rp, I], "mmd_atom_counts": [P, ip, ip, ip], "mmd_atom_pbc": [P], "mmd_atom_sort": [P],
        "mmd_neighbor_setup": [P, ip, creal, I, I, I], "mmd_neighbor_build": [P],
        "mmd_neighbor_geometry": [P, ip, ip, ip, ip],
        "mmd_neighbor_info": [P, ip, ip, C.POINTER(C.c_longlong), ip], "mmd_neighbor_tile_stats": [P, C.POINTER(C.c_longlong)], "mmd_neighbor_tile_histogram": [P, I, I, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
        "mmd_neighbor_tile_rows": [P, I, C.POINTER(C.c_ushort), I, ip, ip, ip, I, ip], "mmd_neighbor_download": [P, ip, I, ip],
        "mmd_neighbor_upload": [P, ip, I, ip, I],
        "mmd_force_lj_setup": [P, I, rp, rp, rp],
        "mmd_force_eam_setup": [P, I, I, I, I, I, creal, creal, rp, rp, rp, rp],
        "mmd_force_compute": [P, I, dp, dp], "mmd_force_eam_download_fp": [P, rp],
        "mmd_comm_setup": [P, creal, I, I], "mmd_comm_info": [P, ip, ip, ip, ip, ip],
        "mmd_comm_swap_info": [P, I, dp, ip, ip, ip], "mmd_comm_unique_id": [C.c_char_p],
        "mmd_comm_init_rccl": [P, C.c_char_p, I, I], "mmd_comm_set_host_transport": [P, P, P, P],
        "mmd_comm_transport_info": [P, ip, ip, ip],
        "mmd_comm_exchange": [P], "mmd_comm_borders": [P], "mmd_comm_communicate": [P],
        "mmd_comm_reverse_communicate": [P], "mmd_comm_download_lists": [P, I, ip],
        "mmd_integrate_setup": [P, creal, creal, I, I], "mmd_integrate_initial": [P], "mmd_integrate_final": [P],
        "mmd_thermo_temperature": [P, dp], "mmd_integrate_mark_positions": [P], "mmd_integrate_max_move": [P, dp], "mmd_integrate_run": [P, I, I, I, P, P],
        "mmd_timers": [P, dp, dp, ip], "mmd_run_stats": [P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], "mmd_profile_kernel": [P, I, I, dp], "mmd_set_option": [P, C.c_char_p, I],
        "mmd_get_counter": [P, C.c_char_p, C.POINTER(C.c_longlong)], "mmd_force_eam_set_fp_halo": [P, P, P], "mmd_atom_upload_x": [P, rp, I],
        "mmd_sync": [P],
        "mmd_input_read": [P, C.c_char_p], "mmd_create_box": [I, I, I, D, rp],
        "mmd_create_atoms": [I, I, I, D, rp, rp, I, rp, rp, ip, ip, ip],
        "mmd_lammps_data_read": [C.c_char_p, ip, rp, rp, rp, rp],
        "mmd_lammps_data_select": [I, rp, rp, rp, rp, I, rp, rp, ip, ip, ip],
        "mmd_eam_tables_from_file": [C.c_char_p, I, ip, ip, ip, ip, rp, rp, rp, rp, rp, rp, rp],
        "mmd_sim_set_unique_id": [C.c_char_p], "mmd_sim_set_host_transport": [P, P, P], "mmd_sim_create": [I, C.POINTER(C.c_char_p), I, C.POINTER(P)],
        "mmd_sim_initial": [P], "mmd_sim_run": [P], "mmd_sim_run_steps": [P, I, dp], "mmd_sim_print_perf": [P],
        "mmd_device_count": [], "mmd_launch_env": [ip, ip, ip, ip, C.c_char_p, I], "mmd_launch_rendezvous": [C.c_char_p, I, ip],
        "mmd_mesh_create": [I, I, C.c_char_p, I, C.POINTER(P)], "mmd_mesh_destroy": [P], "mmd_mesh_allgather": [P, P, I, P],
        "mmd_mesh_allreduce": [P, dp, I], "mmd_mesh_info": [P, ip, ip, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
        "mmd_sim_rows": [P, ip, ip, dp, dp, dp, I], "mmd_sim_output": [P, I], "mmd_sim_wants_yaml": [P, ip], "mmd_sim_natoms": [P], "mmd_sim_destroy": [P],
    }
    for name, args in sig.items():
        fn = getattr(L, name)          # AttributeError here = symbol declared in include/mmd.h is not exported
        fn.argtypes = args
        fn.restype = I
    L.mmd_sim_handle.argtypes = [P]
    L.mmd_sim_handle.restype = P
    L.mmd_mesh_sendrecv.argtypes = [P, P, C.c_longlong, I, P, C.c_longlong, I]
    L.mmd_mesh_sendrecv.restype = C.c_longlong
    L._creal = creal
    L._real = np.float64 if precision == "dp" else np.float32
    L._Input = _input_struct(creal)
    L._symbols = list(sig) + ["mmd_sim_handle", "mmd_last_error", "mmd_variant_string", "mmd_mesh_sendrecv"]
    _LIBS[precision] = L
    return L


THERMO_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double)
SENDRECV_FN = C.CFUNCTYPE(C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class Handle:
    """One device handle = the reference's Atom + Neighbor + Force + Comm + Integrate objects of one rank."""

    def __init__(self, precision="dp", device=-1, _borrowed=None):
        self.L = load_library(precision)
        self.real, self.creal = self.L._real, self.L._creal
        self._own = _borrowed is None
        self._keep = []
        if _borrowed is not None:
            self.h = C.c_void_p(_borrowed)
        else:
            self.h = C.c_void_p()
            self._chk(self.L.mmd_create(device, C.byref(self.h)))

    def _chk(self, rc):
        if rc < 0:
            raise MMDError(self.L.mmd_last_error().decode())
        return rc

    def close(self):
        if self._own and self.h:
            self.L.mmd_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _r(self, a):
        return np.ascontiguousarray(a, self.real)

    def _rp(self, a):
        return a.ctypes.data_as(C.POINTER(self.creal)) if a is not None else None

    @staticmethod
    def _ip(a):
        return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None

    # ---- Atom ------------------------------------------------------------------------------
    def set_box(self, prd, lo=None, hi=None):
        prd = self._r(prd)
        lo = self._r([0, 0, 0] if lo is None else lo)
        hi = self._r(prd if hi is None else hi)
        self._chk(self.L.mmd_atom_set_box(self.h, self._rp(prd), self._rp(lo), self._rp(hi)))

    def get_box(self):
        prd, lo, hi = (np.zeros(3, self.real) for _ in range(3))
        self._chk(self.L.mmd_atom_get_box(self.h, self._rp(prd), self._rp(lo), self._rp(hi)))
        return prd, lo, hi

    def set_mass(self, m):
        self._chk(self.L.mmd_atom_set_mass(self.h, self.creal(m)))

    def upload(self, x, v, type_, tag=None, nlocal=None):
        x = self._r(x).reshape(-1, 3)
        nall = x.shape[0]
        nlocal = nall if nlocal is None else nlocal
        v = None if v is None else self._r(v)
        type_ = np.ascontiguousarray(type_, np.int32)
        tag = None if tag is None else np.ascontiguousarray(tag, np.int32)
        self._chk(self.L.mmd_atom_upload(self.h, self._rp(x), self._rp(v), self._ip(type_), self._ip(tag), nlocal, nall - nlocal))

    def upload_x(self, x):
        x = self._r(x).reshape(-1, 3)
        self._chk(self.L.mmd_atom_upload_x(self.h, self._rp(x), x.shape[0]))

    def counts(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mmd_atom_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def download(self, halfneigh=False):
        nl, ng, _ = self.counts()
        x = np.zeros((nl + ng, 3), self.real)
        v = np.zeros((nl, 3), self.real)
        # the library writes nlocal rows of f for a full-list handle and nlocal+nghost rows for a half-list one
        # (include/mmd.h): the buffer always has room for the larger; `halfneigh` only selects what is returned
        f = np.zeros((nl + ng, 3), self.real)
        t = np.zeros(nl + ng, np.int32)
        tag = np.zeros(nl, np.int32)
        self._chk(self.L.mmd_atom_download(self.h, self._rp(x), self._rp(v), self._rp(f), self._ip(t), self._ip(tag)))
        return {"x": x, "v": v, "f": f if halfneigh else f[:nl], "type": t, "tag": tag, "nlocal": nl, "nghost": ng}

    def upload_f(self, f):
        f = self._r(f).reshape(-1, 3)
        self._chk(self.L.mmd_atom_upload_f(self.h, self._rp(f), f.shape[0]))

    def pbc(self):
        self._chk(self.L.mmd_atom_pbc(self.h))

    def sort(self):
        self._chk(self.L.mmd_atom_sort(self.h))

    # ---- Neighbor --------------------------------------------------------------------------
    def neighbor_setup(self, nbin, cutneigh, halfneigh, ghost_newton, ntypes):
        nb = np.ascontiguousarray(nbin, np.int32)
        self._half = bool(halfneigh)
        self._chk(self.L.mmd_neighbor_setup(self.h, self._ip(nb), self.creal(cutneigh), halfneigh, ghost_newton, ntypes))

    def neighbor_geometry(self):
        a = [np.zeros(3, np.int32) for _ in range(4)]
        self._chk(self.L.mmd_neighbor_geometry(self.h, *[self._ip(q) for q in a]))
        return {"mbin": a[0], "mbinlo": a[1], "nblk": a[2], "reach": a[3]}

    def neighbor_build(self):
        self._chk(self.L.mmd_neighbor_build(self.h))

    def neighbor_info(self):
        m, b, mr = C.c_int(), C.c_int(), C.c_int()
        t = C.c_longlong()
        self._chk(self.L.mmd_neighbor_info(self.h, C.byref(m), C.byref(b), C.byref(t), C.byref(mr)))
        return {"maxneighs": m.value, "mbins": b.value, "total": t.value, "max_row": mr.value}

    def mark_positions(self):
        self._chk(self.L.mmd_integrate_mark_positions(self.h))

    def max_move(self):
        d = C.c_double()
        self._chk(self.L.mmd_integrate_max_move(self.h, C.byref(d)))
        return d.value

    def neighbor_tile_stats(self):
        out = (C.c_longlong * 6)()
        self._chk(self.L.mmd_neighbor_tile_stats(self.h, out))
        keys = ("tiles", "max_candidates", "sum_candidates", "sum_padded_rows", "sum_atoms", "max_padded_row")
        return dict(zip(keys, [int(v) for v in out]))

    def neighbor_tile_histogram(self, nb=64, width=16):
        a, b = (C.c_longlong * nb)(), (C.c_longlong * nb)()
        self._chk(self.L.mmd_neighbor_tile_histogram(self.h, nb, width, a, b))
        return [int(v) for v in a], [int(v) for v in b]

    def neighbor_tile_rows(self, tile, rows_cap=256, cand_cap=4096):
        """diagnostic: (rows[kmax][64] of 16-bit LDS record offsets, atoms of the 64 lanes (-1: none), candidate union) of one tile"""
        rows = np.zeros(rows_cap * 64, np.uint16)
        atoms = np.zeros(64, np.int32)
        cand = np.zeros(cand_cap, np.int32)
        km, nc = C.c_int(), C.c_int()
        self._chk(self.L.mmd_neighbor_tile_rows(self.h, tile, rows.ctypes.data_as(C.POINTER(C.c_ushort)), rows_cap * 64, C.byref(km),
                                                atoms.ctypes.data_as(C.POINTER(C.c_int)), cand.ctypes.data_as(C.POINTER(C.c_int)), cand_cap, C.byref(nc)))
        return rows[:km.value * 64].reshape(km.value, 64).copy(), atoms, cand[:nc.value].copy()

    def neighbor_download(self):
        nl = self.counts()[0]
        nn = np.zeros(nl, np.int32)
        # counts first: a half list comes back in the reference's partition (j > i), whose rows differ in length from the device's
        self._chk(self.L.mmd_neighbor_download(self.h, None, 0, self._ip(nn)))
        stride = max(int(nn.max()) if nl else 0, 1)
        nb = np.zeros((nl, stride), np.int32)
        self._chk(self.L.mmd_neighbor_download(self.h, self._ip(nb), stride, self._ip(nn)))
        return nb, nn

    def neighbor_upload(self, neighbors, numneigh):
        nb = np.ascontiguousarray(neighbors, np.int32)
        nn = np.ascontiguousarray(numneigh, np.int32)
        self._chk(self.L.mmd_neighbor_upload(self.h, self._ip(nb), nb.shape[1], self._ip(nn), len(nn)))

    # ---- Force -----------------------------------------------------------------------------
    def force_lj_setup(self, cutforcesq, sigma6, epsilon):
        a, b, c = self._r(cutforcesq), self._r(sigma6), self._r(epsilon)
        nt = int(round(len(a) ** 0.5))
        self._chk(self.L.mmd_force_lj_setup(self.h, nt, self._rp(a), self._rp(b), self._rp(c)))

    def force_eam_setup(self, ntypes, tables):
        t = tables
        a, b, c, d = self._r(t["rhor_spline"]), self._r(t["frho_spline"]), self._r(t["z2r_spline"]), self._r(t["cutforcesq"])
        self._chk(self.L.mmd_force_eam_setup(self.h, ntypes, t["nr"], t["nrho"], t["nr_tot"], t["nrho_tot"], self.creal(t["rdr"]),
                                             self.creal(t["rdrho"]), self._rp(a), self._rp(b), self._rp(c), self._rp(d)))

    def force_compute(self, evflag=1):
        e, v = C.c_double(), C.c_double()
        self._chk(self.L.mmd_force_compute(self.h, evflag, C.byref(e), C.byref(v)))
        return e.value, v.value

    def set_fp_halo(self, fn):
        """install (fn(fp: np.ndarray[nlocal+nghost], nlocal, nghost) -> None, fills fp[nlocal:]) or remove (None) the caller's ForceEAM::communicate"""
        if fn is None:
            self._fp_cb = None
            self._chk(self.L.mmd_force_eam_set_fp_halo(self.h, None, None))
            return
        creal = self.L._creal

        def _cb(ctx, buf, nlocal, nghost):
            try:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(creal)), shape=(nlocal + nghost,))
                fn(arr, nlocal, nghost)
                return 0
            except Exception:  # noqa: BLE001
                return -1
        self._fp_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)(_cb)      # keep the trampoline alive
        self._chk(self.L.mmd_force_eam_set_fp_halo(self.h, C.cast(self._fp_cb, C.c_void_p), None))

    def eam_fp(self):
        nl, ng, _ = self.counts()
        fp = np.zeros(nl + ng, self.real)
        self._chk(self.L.mmd_force_eam_download_fp(self.h, self._rp(fp)))
        return fp

    # ---- Comm ------------------------------------------------------------------------------
    def comm_setup(self, cutneigh, me=0, nprocs=1):
        self._chk(self.L.mmd_comm_setup(self.h, self.creal(cutneigh), me, nprocs))

    def comm_info(self):
        pg, ml, pn, nd = np.zeros(3, np.int32), np.zeros(3, np.int32), np.zeros(6, np.int32), np.zeros(3, np.int32)
        ns = C.c_int()
        self._chk(self.L.mmd_comm_info(self.h, self._ip(pg), self._ip(ml), self._ip(pn), self._ip(nd), C.byref(ns)))
        return {"procgrid": pg, "myloc": ml, "procneigh": pn.reshape(3, 2), "need": nd, "nswap": ns.value}

    def swap_info(self, s):
        slab = (C.c_double * 2)()
        pbc, procs, cnt = (C.c_int * 4)(), (C.c_int * 2)(), (C.c_int * 3)()
        self._chk(self.L.mmd_comm_swap_info(self.h, s, slab, pbc, procs, cnt))
        return {"slablo": slab[0], "slabhi": slab[1], "pbc_any": pbc[0], "pbc": [pbc[1], pbc[2], pbc[3]], "sendproc": procs[0],
                "recvproc": procs[1], "sendnum": cnt[0], "recvnum": cnt[1], "firstrecv": cnt[2]}

    def sendlist(self, s):
        n = self.swap_info(s)["sendnum"]
        out = np.zeros(max(n, 1), np.int32)
        self._chk(self.L.mmd_comm_download_lists(self.h, s, self._ip(out)))
        return out[:n]

    def set_host_transport(self, sendrecv, allreduce):
        """sendrecv(send_bytes, dest, nrecv, src) -> bytes ; allreduce(np.ndarray[float64]) in place"""
        def _sr(ctx, sbuf, ns, dest, rbuf, nr, src):
            data = C.string_at(sbuf, ns) if ns else b""
            got = sendrecv(data, dest, nr, src)
            if got:
                C.memmove(rbuf, got, len(got))
            return len(got)

        def _ar(ctx, vals, n):
            a = np.ctypeslib.as_array(vals, shape=(n,))
            allreduce(a)
            return 0
        cb1, cb2 = SENDRECV_FN(_sr), ALLREDUCE_FN(_ar)
        self._keep += [cb1, cb2]
        self._chk(self.L.mmd_comm_set_host_transport(self.h, C.cast(cb1, C.c_void_p), C.cast(cb2, C.c_void_p), None))

    def init_rccl(self, unique_id, rank, nranks):
        self._chk(self.L.mmd_comm_init_rccl(self.h, unique_id, rank, nranks))

    def transport_info(self):
        k, n, r = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.mmd_comm_transport_info(self.h, C.byref(k), C.byref(n), C.byref(r)))
        return {"kind": {0: "none", 1: "rccl", 2: "host"}[k.value], "nranks": n.value, "rank": r.value}

    def unique_id(self):
        buf = C.create_string_buffer(128)
        self._chk(self.L.mmd_comm_unique_id(buf))
        return buf.raw

    def exchange(self):
        self._chk(self.L.mmd_comm_exchange(self.h))

    def borders(self):
        self._chk(self.L.mmd_comm_borders(self.h))

    def communicate(self):
        self._chk(self.L.mmd_comm_communicate(self.h))

    def reverse_communicate(self):
        self._chk(self.L.mmd_comm_reverse_communicate(self.h))

    # ---- Integrate / Thermo ----------------------------------------------------------------
    def integrate_setup(self, dt, dtforce, neigh_every=20, sort_every=20):
        self._chk(self.L.mmd_integrate_setup(self.h, self.creal(dt), self.creal(dtforce), neigh_every, sort_every))

    def initial_integrate(self):
        self._chk(self.L.mmd_integrate_initial(self.h))

    def final_integrate(self):
        self._chk(self.L.mmd_integrate_final(self.h))

    def temperature_sum(self):
        t = C.c_double()
        self._chk(self.L.mmd_thermo_temperature(self.h, C.byref(t)))
        return t.value

    def run(self, ntimes, thermo_nstat=0, first_step=0):
        rows = []
        cb = THERMO_FN(lambda ctx, step, mv2, eng, vir: rows.append((step, mv2, eng, vir)))
        self._chk(self.L.mmd_integrate_run(self.h, first_step, ntimes, thermo_nstat, C.cast(cb, C.c_void_p), None))
        return rows

    def timers(self):
        t = (C.c_double * 5)()
        ms, n = C.c_double(), C.c_int()
        self._chk(self.L.mmd_timers(self.h, t, C.byref(ms), C.byref(n)))
        return {"total": t[0], "comm": t[1], "force": t[2], "neigh": t[3], "extra": t[4], "force_kernel_ms": ms.value, "force_launches": n.value}

    def run_stats(self):
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._chk(self.L.mmd_run_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"host_syncs": a.value, "bytes_sent": b.value, "transport_syncs": c.value}

    def counter(self, name):
        v = C.c_longlong()
        self._chk(self.L.mmd_get_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    def profile_kernel(self, which, nrep=20):
        ms = C.c_double()
        self._chk(self.L.mmd_profile_kernel(self.h, which, nrep, C.byref(ms)))
        return ms.value

    def set_option(self, name, value):
        self._chk(self.L.mmd_set_option(self.h, name.encode(), int(value)))

    def sync(self):
        self._chk(self.L.mmd_sync(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, gib = C.c_int(), C.c_double()
        self._chk(self.L.mmd_device_info(self.h, name, 256, C.byref(cu), C.byref(gib)))
        return {"name": name.value.decode(), "cus": cu.value, "hbm_gib": gib.value}


# ---- host-side setup helpers (no GPU) --------------------------------------------------------------
def input_read(path, precision="dp"):
    L = load_library(precision)
    inp = L._Input()
    if L.mmd_input_read(C.byref(inp), path.encode()) != 0:
        raise MMDError(L.mmd_last_error().decode())
    return inp


def create_box(nx, ny, nz, rho, precision="dp"):
    L = load_library(precision)
    prd = np.zeros(3, L._real)
    L.mmd_create_box(nx, ny, nz, float(rho), prd.ctypes.data_as(C.POINTER(L._creal)))
    return prd


def create_atoms(nx, ny, nz, rho, lo, hi, ntypes=4, precision="dp"):
    L = load_library(precision)
    rp = C.POINTER(L._creal)
    lo, hi = np.ascontiguousarray(lo, L._real), np.ascontiguousarray(hi, L._real)
    n = C.c_int()
    L.mmd_create_atoms(nx, ny, nz, float(rho), lo.ctypes.data_as(rp), hi.ctypes.data_as(rp), ntypes, None, None, None, None, C.byref(n))
    x, v = np.zeros((n.value, 3), L._real), np.zeros((n.value, 3), L._real)
    t, tag = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
    ipt = C.POINTER(C.c_int)
    L.mmd_create_atoms(nx, ny, nz, float(rho), lo.ctypes.data_as(rp), hi.ctypes.data_as(rp), ntypes, x.ctypes.data_as(rp),
                       v.ctypes.data_as(rp), t.ctypes.data_as(ipt), tag.ctypes.data_as(ipt), C.byref(n))
    return x, v, t, tag


def lammps_data_read(path, precision="dp"):
    """read_lammps_data's parsing part (ref/setup.cpp:55-301): (natoms, prd[3], mass or None, x[natoms,3], v[natoms,3])"""
    L = load_library(precision)
    rp = C.POINTER(L._creal)
    n, mass = C.c_int(), L._creal()
    prd = np.zeros(3, L._real)
    if L.mmd_lammps_data_read(os.fsencode(path), C.byref(n), prd.ctypes.data_as(rp), C.byref(mass), None, None) < 0:
        raise MMDError(L.mmd_last_error().decode())
    x, v = np.zeros((n.value, 3), L._real), np.zeros((n.value, 3), L._real)
    if L.mmd_lammps_data_read(os.fsencode(path), C.byref(n), prd.ctypes.data_as(rp), C.byref(mass), x.ctypes.data_as(rp), v.ctypes.data_as(rp)) < 0:
        raise MMDError(L.mmd_last_error().decode())
    return n.value, prd, (None if mass.value < 0 else mass.value), x, v


def lammps_data_select(x_all, v_all, lo, hi, ntypes=4, precision="dp"):
    """atoms of sub-box [lo,hi) in file order (ref/setup.cpp:281-286): x, v, type, tag(= file id)"""
    L = load_library(precision)
    rp, ipt = C.POINTER(L._creal), C.POINTER(C.c_int)
    xa, va = np.ascontiguousarray(x_all, L._real), np.ascontiguousarray(v_all, L._real)
    lo, hi = np.ascontiguousarray(lo, L._real), np.ascontiguousarray(hi, L._real)
    n = C.c_int()
    args = (len(xa), xa.ctypes.data_as(rp), va.ctypes.data_as(rp), lo.ctypes.data_as(rp), hi.ctypes.data_as(rp), ntypes)
    if L.mmd_lammps_data_select(*args, None, None, None, None, C.byref(n)) < 0:
        raise MMDError(L.mmd_last_error().decode())
    x, v = np.zeros((n.value, 3), L._real), np.zeros((n.value, 3), L._real)
    t, tag = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
    if L.mmd_lammps_data_select(*args, x.ctypes.data_as(rp), v.ctypes.data_as(rp), t.ctypes.data_as(ipt), tag.ctypes.data_as(ipt), C.byref(n)) < 0:
        raise MMDError(L.mmd_last_error().decode())
    return x, v, t, tag


def eam_tables_from_file(path, ntypes=4, precision="dp"):
    L = load_library(precision)
    rp = C.POINTER(L._creal)
    nr, nrho, nrt, nrhot = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rdr, rdrho, cut, mass = L._creal(), L._creal(), L._creal(), L._creal()
    rc = L.mmd_eam_tables_from_file(path.encode(), ntypes, C.byref(nr), C.byref(nrho), C.byref(nrt), C.byref(nrhot), C.byref(rdr),
                                    C.byref(rdrho), C.byref(cut), C.byref(mass), None, None, None)
    if rc < 0:
        raise MMDError(L.mmd_last_error().decode())
    n2 = ntypes * ntypes
    a, b, c = np.zeros(n2 * nrt.value, L._real), np.zeros(n2 * nrhot.value, L._real), np.zeros(n2 * nrt.value, L._real)
    L.mmd_eam_tables_from_file(path.encode(), ntypes, C.byref(nr), C.byref(nrho), C.byref(nrt), C.byref(nrhot), C.byref(rdr),
                               C.byref(rdrho), C.byref(cut), C.byref(mass), a.ctypes.data_as(rp), b.ctypes.data_as(rp), c.ctypes.data_as(rp))
    return {"nr": nr.value, "nrho": nrho.value, "nr_tot": nrt.value, "nrho_tot": nrhot.value, "rdr": rdr.value, "rdrho": rdrho.value,
            "cutmax": cut.value, "mass": mass.value, "rhor_spline": a, "frho_spline": b, "z2r_spline": c,
            "cutforcesq": np.full(n2, L._real(cut.value) * L._real(cut.value), L._real)}


def launch_env(precision="dp"):
    """rank / size / local rank as the launcher's environment describes them (torchrun, Open MPI, MPICH hydra, PMIx, Slurm: include/mmd.h)"""
    L = load_library(precision)
    r, n, lr, ls = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(32)
    if L.mmd_launch_env(C.byref(r), C.byref(n), C.byref(lr), C.byref(ls), name, 32) < 0:
        raise MMDError(L.mmd_last_error().decode())
    return {"rank": r.value, "nranks": n.value, "local_rank": lr.value, "local_size": ls.value, "launcher": name.value.decode()}


def launch_rendezvous(precision="dp"):
    L = load_library(precision)
    addr = C.create_string_buffer(256)
    port = C.c_int()
    if L.mmd_launch_rendezvous(addr, 256, C.byref(port)) < 0:
        raise MMDError(L.mmd_last_error().decode())
    return addr.value.decode(), port.value


class Mesh:
    """the executable's built-in TCP mesh (csrc/launch.cpp): rendezvous of the ranks and host-staged debug transport"""

    def __init__(self, rank, nranks, addr="127.0.0.1", port=29500, precision="dp"):
        self.L = load_library(precision)
        self.p = C.c_void_p()
        if self.L.mmd_mesh_create(rank, nranks, addr.encode(), port, C.byref(self.p)) < 0:
            raise MMDError(self.L.mmd_last_error().decode())
        self.rank, self.nranks = rank, nranks

    def sendrecv(self, data: bytes, dest: int, nrecv: int, src: int) -> bytes:
        rbuf = C.create_string_buffer(max(nrecv, 1))
        got = self.L.mmd_mesh_sendrecv(self.p, data, len(data), dest, rbuf, nrecv, src)
        if got < 0:
            raise MMDError(self.L.mmd_last_error().decode())
        return rbuf.raw[:got]

    def allreduce(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if self.L.mmd_mesh_allreduce(self.p, a.ctypes.data_as(C.POINTER(C.c_double)), a.size) < 0:
            raise MMDError(self.L.mmd_last_error().decode())
        arr[:] = a

    def allgather(self, mine: bytes) -> list:
        out = C.create_string_buffer(len(mine) * self.nranks)
        if self.L.mmd_mesh_allgather(self.p, mine, len(mine), out) < 0:
            raise MMDError(self.L.mmd_last_error().decode())
        return [out.raw[i * len(mine):(i + 1) * len(mine)] for i in range(self.nranks)]

    def info(self):
        r, n, b, m = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
        self.L.mmd_mesh_info(self.p, C.byref(r), C.byref(n), C.byref(b), C.byref(m))
        return {"rank": r.value, "nranks": n.value, "bytes_sent": b.value, "messages": m.value}

    def close(self):
        if self.p:
            self.L.mmd_mesh_destroy(self.p)
            self.p = C.c_void_p()


_SIM_TRANSPORT_KEEP = []


def sim_set_host_transport(sendrecv, allreduce, precision="dp"):
    """register a host-staged transport (e.g. minimd_amd.transport.GlooTransport) for subsequently created Sims"""
    L = load_library(precision)

    def _sr(ctx, sbuf, ns, dest, rbuf, nr, src):
        data = C.string_at(sbuf, ns) if ns else b""
        got = sendrecv(data, dest, nr, src)
        if got:
            C.memmove(rbuf, got, len(got))
        return len(got)

    def _ar(ctx, vals, n):
        a = np.ctypeslib.as_array(vals, shape=(n,))
        allreduce(a)
        return 0
    cb1, cb2 = SENDRECV_FN(_sr), ALLREDUCE_FN(_ar)
    _SIM_TRANSPORT_KEEP.extend([cb1, cb2])
    L.mmd_sim_set_host_transport(C.cast(cb1, C.c_void_p), C.cast(cb2, C.c_void_p), None)


class Sim:
    """twin of the reference executable's main(): Sim(["-i", deck, "-s", "32", ...]).initial(); .run()"""

    def __init__(self, args, precision="dp", quiet=True, cwd=None):
        self.L = load_library(precision)
        args = [str(a) for a in args]
        if "-i" in args or "--input_file" in args:
            k = args.index("-i") if "-i" in args else args.index("--input_file")
            if not os.path.isabs(args[k + 1]):
                args[k + 1] = os.path.join(DATA_DIR, args[k + 1])
        else:
            args = ["-i", os.path.join(DATA_DIR, "in.lj.miniMD")] + args
        argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
        self.s = C.c_void_p()
        old = os.getcwd()
        os.chdir(cwd or DATA_DIR)          # Cu_u6.eam is looked up in the CWD, as in the reference
        try:
            rc = self.L.mmd_sim_create(len(args), argv, 1 if quiet else 0, C.byref(self.s))
        finally:
            os.chdir(old)
        if rc != 0:
            raise MMDError(self.L.mmd_last_error().decode())
        self.handle = Handle(precision, _borrowed=self.L.mmd_sim_handle(self.s))

    def _chk(self, rc):
        if rc < 0:
            raise MMDError(self.L.mmd_last_error().decode())

    def initial(self):
        self._chk(self.L.mmd_sim_initial(self.s))

    def run(self):
        self._chk(self.L.mmd_sim_run(self.s))

    def run_steps(self, n):
        sec = C.c_double()
        self._chk(self.L.mmd_sim_run_steps(self.s, n, C.byref(sec)))
        return sec.value

    def print_perf(self):
        self.L.mmd_sim_print_perf(self.s)

    def natoms(self):
        return self.L.mmd_sim_natoms(self.s)

    def rows(self):
        n = C.c_int()
        self.L.mmd_sim_rows(self.s, C.byref(n), None, None, None, None, 0)
        m = max(n.value, 1)
        st = (C.c_int * m)()
        t, u, p = (C.c_double * m)(), (C.c_double * m)(), (C.c_double * m)()
        self.L.mmd_sim_rows(self.s, C.byref(n), st, t, u, p, m)
        return [(st[i], t[i], u[i], p[i]) for i in range(n.value)]

    def close(self):
        if self.s:
            self.L.mmd_sim_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
