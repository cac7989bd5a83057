// minimd_amd/csrc/comm.hip — Comm (ref/comm.cpp): spatial decomposition, ghost-atom halo (communicate /
// reverse_communicate), atom migration (exchange) and ghost-list construction (borders) with all per-atom
// work in HIP kernels (ordered stream compaction, pack, unpack) and only counts crossing to the host.
//
// Transport between ranks is RCCL point-to-point (ncclSend/ncclRecv grouped per swap, xGMI) on the
// handle's stream; a host-staged callback transport exists for tests (gloo). Swaps whose partner is this
// rank (periodic images on a 1-wide processor grid) never touch a buffer: one kernel reads the source
// atoms and writes the ghosts in place.
//
// Halo element = real4 {x,y,z,(real)type} so a remote swap is received straight into x[firstrecv...]
// (unpack elided).
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <thread>

#include "device_utils.hpp"
#include "mmd_internal.hpp"
#include "tile_lds.hpp"

#define NCCL_TRY(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if(_r != ncclSuccess) {                                                                     \
      mmd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return -1;                                                                                \
    }                                                                                           \
  } while(0)

// ---------------------------------------------------------------------------------------------------
// Comm::setup (ref/comm.cpp:60-272) — host only
// ---------------------------------------------------------------------------------------------------
static int cart_rank(const int pg[3], int c0, int c1, int c2)
{
  // MPI_Cart_create(reorder = 0) numbering: row-major over (x,y,z), periodic wrap
  c0 = (c0 % pg[0] + pg[0]) % pg[0];
  c1 = (c1 % pg[1] + pg[1]) % pg[1];
  c2 = (c2 % pg[2] + pg[2]) % pg[2];
  return (c0 * pg[1] + c1) * pg[2] + c2;
}

extern "C" int mmd_comm_setup(mmd_handle* h, mmd_float cutneigh, int me, int nprocs)
{
  if(!h || nprocs < 1 || me < 0 || me >= nprocs) { mmd_set_error("mmd_comm_setup: bad arguments"); return -1; }
  if(!(h->prd[0] > 0)) { mmd_set_error("mmd_comm_setup: box not set"); return -1; }
  h->me = me;
  h->nprocs = nprocs;
  const real* prd = h->prd;
  const real area[3] = {prd[0] * prd[1], prd[0] * prd[2], prd[1] * prd[2]};
  real bestsurf = 2.0 * (area[0] + area[1] + area[2]);
  int pg[3] = {0, 0, 0};
  for(int ipx = 1; ipx <= nprocs; ipx++) {
    if(nprocs % ipx) continue;
    const int nremain = nprocs / ipx;
    for(int ipy = 1; ipy <= nremain; ipy++) {
      if(nremain % ipy) continue;
      const int ipz = nremain / ipy;
      const real surf = area[0] / ipx / ipy + area[1] / ipx / ipz + area[2] / ipy / ipz;
      if(surf < bestsurf) { bestsurf = surf; pg[0] = ipx; pg[1] = ipy; pg[2] = ipz; }
    }
  }
  if(pg[0] * pg[1] * pg[2] != nprocs) { mmd_set_error("ERROR: Bad grid of processors"); return -1; }
  for(int d = 0; d < 3; d++) h->procgrid[d] = pg[d];
  h->myloc[0] = me / (pg[1] * pg[2]);
  h->myloc[1] = (me / pg[2]) % pg[1];
  h->myloc[2] = me % pg[2];
  for(int d = 0; d < 3; d++) {
    int lo[3] = {h->myloc[0], h->myloc[1], h->myloc[2]}, hi[3] = {h->myloc[0], h->myloc[1], h->myloc[2]};
    lo[d] -= 1; hi[d] += 1;
    h->procneigh[d][0] = cart_rank(pg, lo[0], lo[1], lo[2]);
    h->procneigh[d][1] = cart_rank(pg, hi[0], hi[1], hi[2]);
    h->lo[d] = h->myloc[d] * prd[d] / pg[d];
    h->hi[d] = (h->myloc[d] + 1) * prd[d] / pg[d];
    h->need[d] = static_cast<int>(cutneigh * pg[d] / prd[d] + 1);
  }
  for(int d = 0; d < 3; d++) { h->bg.sublo[d] = h->bg_ref.sublo[d] = h->lo[d]; h->bg.subhi[d] = h->bg_ref.subhi[d] = h->hi[d]; }      // (a Neighbor::setup that ran before this)
  for(auto& s : h->swaps) s.sendlist.release();
  h->swaps.clear();
  // a set-up on a handle that has run before: nothing sized from the previous decomposition may survive it — the fixed-size messages of the direct
  // borders / exchange are derived from the PREVIOUS counts on both sides of a pair, and the overlap choice was measured on the old grid
  h->dh.prev_valid = false; h->dh.gmap_live = false; h->dh.ready = false; h->borders_general_done = false; h->ex_prev_valid = false; h->overlap_choice = -1; h->trial_armed = false;
  for(int d = 0; d < 3; d++) {
    for(int ineed = 0; ineed < 2 * h->need[d]; ineed++) {
      Swap s;
      s.dim = d;
      s.pbc_any = 0; s.pbc[0] = s.pbc[1] = s.pbc[2] = 0;
      real lo, hi;
      if(ineed % 2 == 0) {
        s.sendproc = h->procneigh[d][0]; s.recvproc = h->procneigh[d][1];
        const int nbox = h->myloc[d] + ineed / 2;
        lo = nbox * prd[d] / pg[d];
        hi = h->lo[d] + cutneigh;
        const real cap = (nbox + 1) * prd[d] / pg[d];
        hi = hi < cap ? hi : cap;
        if(h->myloc[d] == 0) { s.pbc_any = 1; s.pbc[d] = 1; }
      } else {
        s.sendproc = h->procneigh[d][1]; s.recvproc = h->procneigh[d][0];
        const int nbox = h->myloc[d] - ineed / 2;
        hi = (nbox + 1) * prd[d] / pg[d];
        lo = h->hi[d] - cutneigh;
        const real floor_ = nbox * prd[d] / pg[d];
        lo = lo > floor_ ? lo : floor_;
        if(h->myloc[d] == pg[d] - 1) { s.pbc_any = 1; s.pbc[d] = -1; }
      }
      s.slablo = lo; s.slabhi = hi;
      h->swaps.push_back(std::move(s));
    }
  }
  return 0;
}

extern "C" int mmd_comm_info(mmd_handle* h, int procgrid[3], int myloc[3], int procneigh[6], int need[3], int* nswap)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  for(int d = 0; d < 3; d++) {
    if(procgrid) procgrid[d] = h->procgrid[d];
    if(myloc) myloc[d] = h->myloc[d];
    if(need) need[d] = h->need[d];
    if(procneigh) { procneigh[2 * d] = h->procneigh[d][0]; procneigh[2 * d + 1] = h->procneigh[d][1]; }
  }
  if(nswap) *nswap = (int)h->swaps.size();
  return 0;
}

extern "C" int mmd_comm_swap_info(mmd_handle* h, int iswap, double slab[2], int pbc[4], int procs[2], int counts[3])
{
  if(!h || iswap < 0 || iswap >= (int)h->swaps.size()) { mmd_set_error("mmd_comm_swap_info: bad swap index"); return -1; }
  const Swap& s = h->swaps[iswap];
  if(slab) { slab[0] = s.slablo; slab[1] = s.slabhi; }
  if(pbc) { pbc[0] = s.pbc_any; pbc[1] = s.pbc[0]; pbc[2] = s.pbc[1]; pbc[3] = s.pbc[2]; }
  if(procs) { procs[0] = s.sendproc; procs[1] = s.recvproc; }
  if(counts) { counts[0] = s.sendnum; counts[1] = s.recvnum; counts[2] = s.firstrecv; }
  return 0;
}

extern "C" int mmd_comm_download_lists(mmd_handle* h, int iswap, int* sendlist)
{
  if(!h || iswap < 0 || iswap >= (int)h->swaps.size() || !sendlist) { mmd_set_error("mmd_comm_download_lists: bad arguments"); return -1; }
  MMD_TRY(mmd_comm_sendlists_ensure(h));
  const Swap& s = h->swaps[iswap];
  if(s.sendnum) HIP_TRY(hipMemcpyAsync(sendlist, s.sendlist.p, (size_t)s.sendnum * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------------
extern "C" int mmd_comm_unique_id(unsigned char id[128])
{
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, &u, 128);
  return 0;
}

// ---- RCCL bring-up with a diagnosis instead of a hang ---------------------------------------------------------------------------------------------
// The first multi-GPU lease is the first time ncclSend / ncclRecv run between two devices. A mis-wired node (a rank without its device, a dead xGMI link, a rank
// that never arrives) shows as a process that sits in ncclCommInitRank or in its first grouped send/recv until somebody's time limit ends the lease. So:
//  * ncclCommInitRank runs on a helper thread with a bounded wait (MMD_RCCL_TIMEOUT seconds, default 90);
//  * right behind it ONE grouped send/recv of a known pattern with every distinct partner of Comm::setup's grid (the up to 26 neighbours of the direct halo) and
//    ONE all-reduce, both verified, both with the same bounded wait on the stream;
//  * every failure names rank, device (PCI bus id), the partners and what was seen, on stderr of the rank that saw it.
// nranks == 1 (loop-back: bench.py's rank_path_loopback, the tests) runs the same code with itself as the only partner.
namespace {
constexpr int SC_N = 256;                       // ints per partner message
inline int sc_value(int from, int to, int i) { return from * 1000003 + to * 101 + i * 7 + 13; }
double sc_timeout_s()
{
  const char* e = getenv("MMD_RCCL_TIMEOUT");
  const double v = e && *e ? atof(e) : 90.0;
  return v > 0 ? v : 90.0;
}
// the stream drained within the limit?
bool sc_wait(hipStream_t st, double seconds)
{
  const double t0 = mmd_wall();
  for(;;) {
    const hipError_t q = hipStreamQuery(st);
    if(q == hipSuccess) return true;
    if(q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
    if(mmd_wall() - t0 > seconds) { (void)hipGetLastError(); return false; }
    usleep(200);
  }
}
}  // namespace

static int rccl_self_check(mmd_handle* h, ncclComm_t c, int rank, int nranks, const char* where)
{
  // distinct partners: the 26 grid neighbours when Comm::setup has run for this many ranks, else every other rank (up to 26), else the ring neighbours
  std::vector<int> partners;
  auto add = [&](int r) { if(r != rank && std::find(partners.begin(), partners.end(), r) == partners.end()) partners.push_back(r); };
  if(nranks == 1) partners.push_back(0);
  else if(h->nprocs == nranks && h->procgrid[0] * h->procgrid[1] * h->procgrid[2] == nranks) {
    for(int a = -1; a <= 1; a++) for(int b = -1; b <= 1; b++) for(int d = -1; d <= 1; d++)
      add(cart_rank(h->procgrid, h->myloc[0] + a, h->myloc[1] + b, h->myloc[2] + d));
  } else if(nranks <= 27) { for(int r = 0; r < nranks; r++) add(r); }
  else { add((rank + 1) % nranks); add((rank + nranks - 1) % nranks); }
  std::sort(partners.begin(), partners.end());
  const int np = (int)partners.size();
  std::string plist;
  for(int r : partners) plist += (plist.empty() ? "" : ",") + std::to_string(r);
  const double limit = sc_timeout_s();
  const double t0 = mmd_wall();
  int* d_buf = nullptr;
  double* d_red = nullptr;
  HIP_TRY(hipMalloc((void**)&d_buf, (size_t)2 * std::max(np, 1) * SC_N * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&d_red, 2 * sizeof(double)));
  std::vector<int> hb((size_t)2 * std::max(np, 1) * SC_N, -1);
  for(int k = 0; k < np; k++) for(int i = 0; i < SC_N; i++) hb[(size_t)k * SC_N + i] = sc_value(rank, partners[k], i);
  HIP_TRY(hipMemcpyAsync(d_buf, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  auto fail = [&](const char* what) {
    fprintf(stderr, "miniMD-HIP: RCCL bring-up FAILED on rank %d of %d (%s, %s): %s — partners %s; NCCL_DEBUG=INFO shows the transport RCCL chose per peer, "
                    "MMD_TRANSPORT=tcp runs over the host mesh instead, MMD_RCCL_TIMEOUT sets this check's patience (now %g s)\n", rank, nranks, where, h->pci, what, plist.c_str(), limit);
    fflush(stderr);
    mmd_set_error("RCCL bring-up failed on rank %d of %d (%s): %s (partners %s)", rank, nranks, h->pci, what, plist.c_str());
    (void)ncclCommAbort(c);                    // (kills whatever RCCL kernel still waits for its peer: the process can end with the message instead of hanging)
    (void)hipFree(d_buf); (void)hipFree(d_red);
    return -1;
  };
  // (1) one grouped send/recv with every distinct partner: same order on both sides of every pair (ascending partner rank)
  NCCL_TRY(ncclGroupStart());
  for(int k = 0; k < np; k++) {
    NCCL_TRY(ncclSend(d_buf + (size_t)k * SC_N, SC_N, ncclInt, partners[k], c, h->stream));
    NCCL_TRY(ncclRecv(d_buf + (size_t)(np + k) * SC_N, SC_N, ncclInt, partners[k], c, h->stream));
  }
  NCCL_TRY(ncclGroupEnd());
  if(!sc_wait(h->stream, limit)) return fail("the grouped ncclSend/ncclRecv with the partners did not complete (a partner never posted its side, or a link between two of the devices does not carry data)");
  HIP_TRY(hipMemcpy(hb.data(), d_buf, hb.size() * sizeof(int), hipMemcpyDeviceToHost));
  for(int k = 0; k < np; k++)
    for(int i = 0; i < SC_N; i++)
      if(hb[(size_t)(np + k) * SC_N + i] != sc_value(partners[k], rank, i)) {
        char b[200];
        snprintf(b, sizeof(b), "message from rank %d arrived corrupted: word %d is %d, expected %d", partners[k], i, hb[(size_t)(np + k) * SC_N + i], sc_value(partners[k], rank, i));
        return fail(b);
      }
  // (2) one all-reduce (the thermo sums' collective, ref/thermo.cpp:131-133)
  const double mine[2] = {(double)(rank + 1), 1.0};
  HIP_TRY(hipMemcpyAsync(d_red, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
  NCCL_TRY(ncclAllReduce(d_red, d_red, 2, ncclDouble, ncclSum, c, h->stream));
  if(!sc_wait(h->stream, limit)) return fail("ncclAllReduce over all ranks did not complete (some rank is not in the collective)");
  double got[2] = {0, 0};
  HIP_TRY(hipMemcpy(got, d_red, sizeof(got), hipMemcpyDeviceToHost));
  if(got[0] != 0.5 * nranks * (nranks + 1.0) || got[1] != (double)nranks) {
    char b[160];
    snprintf(b, sizeof(b), "ncclAllReduce(sum) returned %.1f / %.1f, expected %.1f / %d", got[0], got[1], 0.5 * nranks * (nranks + 1.0), nranks);
    return fail(b);
  }
  HIP_TRY(hipFree(d_buf));
  HIP_TRY(hipFree(d_red));
  h->rccl_check_partners = np;
  h->rccl_check_s = mmd_wall() - t0;
  return 0;
}

extern "C" int mmd_comm_init_rccl(mmd_handle* h, const unsigned char id[128], int rank, int nranks)
{
  if(!h || !id || nranks < 1 || rank < 0 || rank >= nranks) { mmd_set_error("mmd_comm_init_rccl: bad arguments"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  if(hipDeviceGetPCIBusId(h->pci, (int)sizeof(h->pci), h->device) != hipSuccess) { (void)hipGetLastError(); snprintf(h->pci, sizeof(h->pci), "device %d", h->device); }
  ncclUniqueId u;
  memcpy(&u, id, 128);
  // ncclCommInitRank blocks until every rank of the communicator has arrived: on a helper thread, so that a rank that never comes ends in a message
  struct Init { ncclComm_t c = nullptr; ncclResult_t r = ncclSuccess; std::atomic<int> done{0}; };
  auto st = std::make_shared<Init>();
  const int dev = h->device;
  std::thread([st, u, rank, nranks, dev]() {
    (void)hipSetDevice(dev);
    st->r = ncclCommInitRank(&st->c, nranks, u, rank);
    st->done.store(1, std::memory_order_release);
  }).detach();
  const double limit = sc_timeout_s(), t0 = mmd_wall();
  while(!st->done.load(std::memory_order_acquire)) {
    if(mmd_wall() - t0 > limit) {
      fprintf(stderr, "miniMD-HIP: RCCL bring-up FAILED on rank %d of %d (%s): ncclCommInitRank did not return within %g s — not every rank of the communicator arrived "
                      "(did all %d ranks start? same ncclUniqueId on all of them? one device per rank?); MMD_RCCL_TIMEOUT sets the patience\n", rank, nranks, h->pci, limit, nranks);
      fflush(stderr);
      mmd_set_error("RCCL bring-up: ncclCommInitRank did not return within %g s on rank %d of %d (%s)", limit, rank, nranks, h->pci);
      return -1;
    }
    usleep(500);
  }
  if(st->r != ncclSuccess) {
    fprintf(stderr, "miniMD-HIP: RCCL bring-up FAILED on rank %d of %d (%s): ncclCommInitRank: %s\n", rank, nranks, h->pci, ncclGetErrorString(st->r));
    mmd_set_error("ncclCommInitRank failed on rank %d of %d (%s): %s", rank, nranks, h->pci, ncclGetErrorString(st->r));
    return -1;
  }
  const char* skip = getenv("MMD_RCCL_SELFCHECK");
  if(!(skip && !strcmp(skip, "0"))) MMD_TRY(rccl_self_check(h, st->c, rank, nranks, "first exchange"));
  h->rccl = (void*)st->c;
  return 0;
}

// which transport the halos use: 1 = RCCL (nranks / rank as ncclCommCount / ncclCommUserRank report them), 2 = host-staged
// callbacks, 0 = none (single rank: every swap is a self swap)
extern "C" int mmd_comm_transport_info(mmd_handle* h, int* kind, int* nranks, int* rank)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  int k = 0, n = h->nprocs, r = h->me;
  if(h->rccl) {
    k = 1;
    NCCL_TRY(ncclCommCount((ncclComm_t)h->rccl, &n));
    NCCL_TRY(ncclCommUserRank((ncclComm_t)h->rccl, &r));
  } else if(h->host_sr) k = 2;
  if(kind) *kind = k;
  if(nranks) *nranks = n;
  if(rank) *rank = r;
  return 0;
}

extern "C" int mmd_comm_set_host_transport(mmd_handle* h, mmd_sendrecv_fn sr, mmd_allreduce_fn ar, void* ctx)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  h->host_sr = sr; h->host_ar = ar; h->host_ctx = ctx;
  return 0;
}

// device buffers in, device buffers out; byte counts
int mmd_transport_sendrecv(mmd_handle* h, const void* dsend, size_t nsend, int dest, void* drecv, size_t nrecv, int src)
{
  h->halo_bytes += (long long)nsend;
  if(h->rccl) {
    ncclComm_t c = (ncclComm_t)h->rccl;
    NCCL_TRY(ncclGroupStart());
    if(nsend) NCCL_TRY(ncclSend(dsend, nsend, ncclChar, dest, c, h->stream));
    if(nrecv) NCCL_TRY(ncclRecv(drecv, nrecv, ncclChar, src, c, h->stream));
    NCCL_TRY(ncclGroupEnd());
    return 0;
  }
  if(h->host_sr) {
    if(h->stage_send.size() < nsend + 8) h->stage_send.resize(nsend + 8);
    if(h->stage_recv.size() < nrecv + 8) h->stage_recv.resize(nrecv + 8);
    if(nsend) HIP_TRY(hipMemcpyAsync(h->stage_send.data(), dsend, nsend, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync_transport(h));
    const long long got = h->host_sr(h->host_ctx, h->stage_send.data(), (long long)nsend, dest, h->stage_recv.data(), (long long)nrecv, src);
    if(got != (long long)nrecv) { mmd_set_error("host transport: expected %zu bytes from rank %d, got %lld", nrecv, src, got); return -1; }
    if(nrecv) HIP_TRY(hipMemcpyAsync(drecv, h->stage_recv.data(), nrecv, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mmd_stream_sync_transport(h));
    return 0;
  }
  mmd_set_error("rank %d has a remote partner but no transport is attached (mmd_comm_init_rccl / mmd_comm_set_host_transport)", h->me);
  return -1;
}

int mmd_transport_sendrecv_counts(mmd_handle* h, int nsend, int dest, int* nrecv, int src)
{
  h->halo_bytes += (long long)sizeof(int);
  if(h->rccl) {
    ncclComm_t c = (ncclComm_t)h->rccl;
    h->h_flags[8] = nsend;
    HIP_TRY(hipMemcpyAsync(h->d_flags + 8, h->h_flags + 8, sizeof(int), hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(ncclGroupStart());
    NCCL_TRY(ncclSend(h->d_flags + 8, 1, ncclInt, dest, c, h->stream));
    NCCL_TRY(ncclRecv(h->d_flags + 9, 1, ncclInt, src, c, h->stream));
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipMemcpyAsync(h->h_flags + 9, h->d_flags + 9, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    *nrecv = h->h_flags[9];
    return 0;
  }
  if(h->host_sr) {
    int out = 0;
    const long long got = h->host_sr(h->host_ctx, &nsend, sizeof(int), dest, &out, sizeof(int), src);
    if(got != (long long)sizeof(int)) { mmd_set_error("host transport: count handshake failed"); return -1; }
    *nrecv = out;
    return 0;
  }
  mmd_set_error("rank %d has a remote partner but no transport is attached", h->me);
  return -1;
}

// the two swaps of one ghost layer of a dimension (towards -1 and towards +1) are independent of each other: their count
// handshakes share ONE ncclGroup and ONE host synchronisation, and so do their payloads. Sends/receives to the same peer
// (2-wide grids: both neighbours are the same rank) are matched in the order they are issued, on both sides alike.
int mmd_transport_sendrecv_counts_pair(mmd_handle* h, const int nsend[2], const int dest[2], int nrecv[2], const int src[2])
{
  if(h->rccl) {
    h->halo_bytes += 2 * (long long)sizeof(int);
    ncclComm_t c = (ncclComm_t)h->rccl;
    h->h_flags[8] = nsend[0]; h->h_flags[9] = nsend[1];
    HIP_TRY(hipMemcpyAsync(h->d_flags + 8, h->h_flags + 8, 2 * sizeof(int), hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(ncclGroupStart());
    for(int q = 0; q < 2; q++) {
      NCCL_TRY(ncclSend(h->d_flags + 8 + q, 1, ncclInt, dest[q], c, h->stream));
      NCCL_TRY(ncclRecv(h->d_flags + 10 + q, 1, ncclInt, src[q], c, h->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipMemcpyAsync(h->h_flags + 10, h->d_flags + 10, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    nrecv[0] = h->h_flags[10]; nrecv[1] = h->h_flags[11];
    return 0;
  }
  for(int q = 0; q < 2; q++) MMD_TRY(mmd_transport_sendrecv_counts(h, nsend[q], dest[q], &nrecv[q], src[q]));
  return 0;
}
int mmd_transport_sendrecv_pair(mmd_handle* h, const void* const dsend[2], const size_t nsend[2], const int dest[2], void* const drecv[2],
                                const size_t nrecv[2], const int src[2])
{
  if(h->rccl) {
    h->halo_bytes += (long long)(nsend[0] + nsend[1]);
    ncclComm_t c = (ncclComm_t)h->rccl;
    NCCL_TRY(ncclGroupStart());
    for(int q = 0; q < 2; q++) {
      if(nsend[q]) NCCL_TRY(ncclSend(dsend[q], nsend[q], ncclChar, dest[q], c, h->stream));
      if(nrecv[q]) NCCL_TRY(ncclRecv(drecv[q], nrecv[q], ncclChar, src[q], c, h->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    return 0;
  }
  for(int q = 0; q < 2; q++) MMD_TRY(mmd_transport_sendrecv(h, dsend[q], nsend[q], dest[q], drecv[q], nrecv[q], src[q]));
  return 0;
}

// in-place sum over ranks of vals[0..n) (MPI_Allreduce SUM of the thermo scalars, ref/thermo.cpp:131,168,188)
int mmd_transport_allreduce(mmd_handle* h, double* vals, int n)
{
  if(h->nprocs == 1) return 0;
  if(h->rccl) {
    ncclComm_t c = (ncclComm_t)h->rccl;
    HIP_TRY(hipMemcpyAsync(h->d_result + 8, vals, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(ncclAllReduce(h->d_result + 8, h->d_result + 8, n, ncclDouble, ncclSum, c, h->stream));
    HIP_TRY(hipMemcpyAsync(vals, h->d_result + 8, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    return 0;
  }
  if(h->host_ar) {
    if(h->host_ar(h->host_ctx, vals, n)) { mmd_set_error("host transport: allreduce failed"); return -1; }
    return 0;
  }
  mmd_set_error("no transport attached for allreduce");
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// ordered stream compaction: out[] = ascending indices i in [first, first+n) with pred(i)
// ---------------------------------------------------------------------------------------------------
struct SlabPred {      // Comm::borders selection, closed slab (ref/comm.cpp:776)
  const real4* x; int dim; real lo, hi;
  __device__ bool operator()(int i) const { const real4 p = x[i]; const real c = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); return c >= lo && c <= hi; }
  __device__ int index(int i) const { return i; }
};
// the same selection over a pre-compacted window: virtual index q < nb -> owned boundary atom bnd[q] (ascending),
// q >= nb -> ghost ghost0 + (q - nb); output order == ascending atom index, as a scan of [0, nlocal+nghost) gives
struct BndSlabPred {
  const real4* x; const int* bnd; int nb, ghost0, dim; real lo, hi;
  __device__ int index(int q) const { return q < nb ? bnd[q] : ghost0 + (q - nb); }
  __device__ bool operator()(int q) const { const real4 p = x[index(q)]; const real c = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); return c >= lo && c <= hi; }
};
// owned atoms that lie in ANY send slab (candidates of every swap): scanned once per borders()
struct AnySlabPred {
  const real4* x; int n; real lo[6], hi[6]; int dim[6];
  __device__ int index(int i) const { return i; }
  __device__ bool operator()(int i) const {
    const real4 p = x[i];
    bool in = false;
    for(int s = 0; s < n; s++) { const real c = dim[s] == 0 ? p.x : (dim[s] == 1 ? p.y : p.z); in = in || (c >= lo[s] && c <= hi[s]); }
    return in;
  }
};
struct LeavePred {     // Comm::exchange leavers, half-open box (ref/comm.cpp:440)
  const real4* x; int dim; real lo, hi;
  __device__ bool operator()(int i) const { const real4 p = x[i]; const real c = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); return c < lo || c >= hi; }
  __device__ int index(int i) const { return i; }
};
struct StayPred {
  const real4* x; int dim; real lo, hi;
  __device__ bool operator()(int i) const { const real4 p = x[i]; const real c = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); return !(c < lo || c >= hi); }
  __device__ int index(int i) const { return i; }
};
struct TileFlagPred {  // tiles whose candidate union does (want=1) / does not (want=0) contain a ghost atom
  const int* flag; int want;
  __device__ bool operator()(int t) const { return (flag[t] != 0) == (want != 0); }
  __device__ int index(int t) const { return t; }
};
struct ExchRec {       // Atom::pack_exchange payload (ref/atom.cpp:228-239) + tag
  real x, y, z, w, vx, vy, vz;
  int tag, pad;
};
struct ArrivePred {    // arrivals that fall inside my box in this dimension (ref/comm.cpp:566-571)
  const ExchRec* r; int dim; real lo, hi;
  __device__ bool operator()(int i) const { const real c = dim == 0 ? r[i].x : (dim == 1 ? r[i].y : r[i].z); return c >= lo && c < hi; }
  __device__ int index(int i) const { return i; }
};

#define CP_TILE 1024
template <class Pred>
__global__ __launch_bounds__(256) void k_compact_count(Pred pred, int first, int n, int* __restrict__ tile_counts)
{
  __shared__ int lds[17];
  const int base = blockIdx.x * CP_TILE + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) if(base + k < n) c += pred(first + base + k) ? 1 : 0;
  int tot;
  block_incl_scan(c, lds, &tot);
  if(threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}
template <class Pred>
__global__ __launch_bounds__(256) void k_compact_scatter(Pred pred, int first, int n, const int* __restrict__ tile_offsets,
                                                         int* __restrict__ out)
{
  __shared__ int lds[17];
  const int base = blockIdx.x * CP_TILE + threadIdx.x * 4;
  bool fl[4];
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) { fl[k] = base + k < n && pred(first + base + k); c += fl[k] ? 1 : 0; }
  int tot;
  const int inc = block_incl_scan(c, lds, &tot);
  int pos = tile_offsets[blockIdx.x] + inc - c;
#pragma unroll
  for(int k = 0; k < 4; k++) if(fl[k]) out[pos++] = pred.index(first + base + k);
}

template <class Pred>
static int compact(mmd_handle* h, Pred pred, int first, int n, DevArr<int>& out, int* count)
{
  *count = 0;
  if(n <= 0) return 0;
  const int ntiles = div_up(n, CP_TILE);
  MMD_TRY(h->flag_tmp.ensure((size_t)ntiles + 8, false, h->stream));
  hipLaunchKernelGGL((k_compact_count<Pred>), dim3(ntiles), dim3(256), 0, h->stream, pred, first, n, h->flag_tmp.p);
  int total = 0;
  MMD_TRY(mmd_exclusive_scan(h, h->flag_tmp.p, ntiles, &total));
  if(total) {
    MMD_TRY(out.ensure((size_t)total + 8, false, h->stream));
    hipLaunchKernelGGL((k_compact_scatter<Pred>), dim3(ntiles), dim3(256), 0, h->stream, pred, first, n, h->flag_tmp.p, out.p);
  }
  HIP_TRY(hipGetLastError());
  *count = total;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Comm::communicate (ref/comm.cpp:276-317) — forward halo of positions
// ---------------------------------------------------------------------------------------------------
// pack (+ PBC shift) ; for a self swap `dst` is x + firstrecv: Atom::pack_comm + unpack_comm fused
__global__ __launch_bounds__(256) void k_pack_comm(const real4* __restrict__ x, const int* __restrict__ list, int n,
                                                   real sx, real sy, real sz, int pbc_any, real4* __restrict__ dst)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  real4 p = x[list[i]];
  if(pbc_any) { p.x += sx; p.y += sy; p.z += sz; }     // x + pbc_flag*prd, flag*prd is exact (flag in {-1,0,1})
  dst[i] = p;
}

__global__ void k_ghost_update(real4* __restrict__ x, int nlocal, int nghost, const int* __restrict__ root,
                               const int* __restrict__ image, real xprd, real yprd, real zprd);

extern "C" int mmd_comm_communicate(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(h->ghost_chain_ok && h->opt_fuse && !h->opt_force_transport) {
    if(h->nghost)
      hipLaunchKernelGGL(k_ghost_update, dim3(div_up(h->nghost, 256)), dim3(256), 0, h->stream, h->x.p, h->nlocal, h->nghost,
                         h->ghost_root.p, h->ghost_image.p, h->prd[0], h->prd[1], h->prd[2]);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if(h->dh.ready) return mmd_dh_exchange(h, 0);           // one exchange with the up to 26 neighbours (DirectHalo, mmd_internal.hpp)
  MMD_TRY(mmd_comm_sendlists_ensure(h));
  // swap by swap (later dimensions forward ghosts received by earlier ones); the two swaps of one dimension are
  // independent of each other, so with RCCL they share one ncclGroup: 3 instead of 6 p2p rounds per step
  const size_t nsw = h->swaps.size();
  for(size_t is = 0; is < nsw; is++) {
    Swap& s = h->swaps[is];
    const real sx = s.pbc[0] * h->prd[0], sy = s.pbc[1] * h->prd[1], sz = s.pbc[2] * h->prd[2];
    const bool self = s.sendproc == h->me && !h->opt_force_transport;
    if(self) {
      if(s.sendnum)
        hipLaunchKernelGGL(k_pack_comm, dim3(div_up(s.sendnum, 256)), dim3(256), 0, h->stream, h->x.p, s.sendlist.p, s.sendnum,
                           sx, sy, sz, s.pbc_any, h->x.p + s.firstrecv);
      continue;
    }
    // pair with the next swap when it belongs to the same dimension, is remote too, and RCCL is the transport
    const bool pair = h->rccl && is + 1 < nsw && h->swaps[is + 1].dim == s.dim && h->need[s.dim] == 1 &&
                      !(h->swaps[is + 1].sendproc == h->me && !h->opt_force_transport);
    Swap* s2 = pair ? &h->swaps[is + 1] : nullptr;
    const size_t n1 = (size_t)s.sendnum, n2 = pair ? (size_t)s2->sendnum : 0;
    MMD_TRY(h->buf_send.ensure((size_t)4 * (n1 + n2) + 16, false, h->stream));
    real4* b1 = (real4*)h->buf_send.p;
    real4* b2 = b1 + n1;
    if(n1) hipLaunchKernelGGL(k_pack_comm, dim3(div_up((long long)n1, 256)), dim3(256), 0, h->stream, h->x.p, s.sendlist.p, (int)n1, sx, sy, sz, s.pbc_any, b1);
    if(n2) {
      const real tx = s2->pbc[0] * h->prd[0], ty = s2->pbc[1] * h->prd[1], tz = s2->pbc[2] * h->prd[2];
      hipLaunchKernelGGL(k_pack_comm, dim3(div_up((long long)n2, 256)), dim3(256), 0, h->stream, h->x.p, s2->sendlist.p, (int)n2, tx, ty, tz, s2->pbc_any, b2);
    }
    if(pair) {
      h->halo_bytes += (long long)((n1 + n2) * sizeof(real4));
      ncclComm_t c = (ncclComm_t)h->rccl;
      NCCL_TRY(ncclGroupStart());
      if(n1) NCCL_TRY(ncclSend(b1, n1 * sizeof(real4), ncclChar, s.sendproc, c, h->stream));
      if(s.recvnum) NCCL_TRY(ncclRecv(h->x.p + s.firstrecv, (size_t)s.recvnum * sizeof(real4), ncclChar, s.recvproc, c, h->stream));
      if(n2) NCCL_TRY(ncclSend(b2, n2 * sizeof(real4), ncclChar, s2->sendproc, c, h->stream));
      if(s2->recvnum) NCCL_TRY(ncclRecv(h->x.p + s2->firstrecv, (size_t)s2->recvnum * sizeof(real4), ncclChar, s2->recvproc, c, h->stream));
      NCCL_TRY(ncclGroupEnd());
      is++;
    } else {
      MMD_TRY(mmd_transport_sendrecv(h, b1, n1 * sizeof(real4), s.sendproc, h->x.p + s.firstrecv, (size_t)s.recvnum * sizeof(real4), s.recvproc));
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Comm::reverse_communicate (ref/comm.cpp:321-355) — ghost forces back to their owners
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack_reverse(real* __restrict__ f, const int* __restrict__ list, int n,
                                                        const real* __restrict__ src)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  const int j = list[i];
  f[3 * (size_t)j + 0] += src[3 * (size_t)i + 0];
  f[3 * (size_t)j + 1] += src[3 * (size_t)i + 1];
  f[3 * (size_t)j + 2] += src[3 * (size_t)i + 2];
}

extern "C" int mmd_comm_reverse_communicate(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  MMD_TRY(mmd_comm_sendlists_ensure(h));
  for(int is = (int)h->swaps.size() - 1; is >= 0; is--) {
    Swap& s = h->swaps[is];
    const real* ghost_f = h->f.p + 3 * (size_t)s.firstrecv;     // Atom::pack_reverse is a contiguous slice
    if(s.sendproc == h->me && !h->opt_force_transport) {
      if(s.sendnum) hipLaunchKernelGGL(k_unpack_reverse, dim3(div_up(s.sendnum, 256)), dim3(256), 0, h->stream, h->f.p, s.sendlist.p, s.sendnum, ghost_f);
    } else {
      MMD_TRY(h->buf_recv.ensure((size_t)3 * s.sendnum + 8, false, h->stream));
      MMD_TRY(mmd_transport_sendrecv(h, ghost_f, (size_t)3 * s.recvnum * sizeof(real), s.recvproc, h->buf_recv.p,
                                     (size_t)3 * s.sendnum * sizeof(real), s.sendproc));
      if(s.sendnum) hipLaunchKernelGGL(k_unpack_reverse, dim3(div_up(s.sendnum, 256)), dim3(256), 0, h->stream, h->f.p, s.sendlist.p, s.sendnum, h->buf_recv.p);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Comm::exchange (ref/comm.cpp:364-597) — PBC wrap + migration of atoms that left the sub-box
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_exchange(const real4* __restrict__ x, const real* __restrict__ v, const int* __restrict__ tag,
                                                       const int* __restrict__ leavers, int n, ExchRec* __restrict__ out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const int i = leavers[k];
  const real4 p = x[i];
  ExchRec r;
  r.x = p.x; r.y = p.y; r.z = p.z; r.w = p.w;
  r.vx = v[3 * (size_t)i + 0]; r.vy = v[3 * (size_t)i + 1]; r.vz = v[3 * (size_t)i + 2];
  r.tag = tag[i]; r.pad = 0;
  out[k] = r;
}
// the k-th hole (leaver below the new end) takes the k-th stayer of the tail: Atom::copy (ref/comm.cpp:491-509)
__global__ __launch_bounds__(256) void k_fill_holes(real4* __restrict__ x, real* __restrict__ v, int* __restrict__ type, int* __restrict__ tag,
                                                    const int* __restrict__ holes, const int* __restrict__ fillers, int n)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const int dst = holes[k], src = fillers[k];
  x[dst] = x[src];
  v[3 * (size_t)dst + 0] = v[3 * (size_t)src + 0]; v[3 * (size_t)dst + 1] = v[3 * (size_t)src + 1]; v[3 * (size_t)dst + 2] = v[3 * (size_t)src + 2];
  type[dst] = type[src];
  tag[dst] = tag[src];
}
__global__ __launch_bounds__(256) void k_unpack_exchange(const ExchRec* __restrict__ rec, const int* __restrict__ keep, int n, int first,
                                                         real4* __restrict__ x, real* __restrict__ v, int* __restrict__ type, int* __restrict__ tag)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const ExchRec r = rec[keep[k]];
  const int i = first + k;
  x[i] = real4{r.x, r.y, r.z, r.w};
  v[3 * (size_t)i + 0] = r.vx; v[3 * (size_t)i + 1] = r.vy; v[3 * (size_t)i + 2] = r.vz;
  type[i] = (int)r.w;
  tag[i] = r.tag;
}

// ---------------------------------------------------------------------------------------------------
// Comm::exchange on several ranks without count handshakes: the leavers of a dimension travel in ONE message of fixed size — a
// 64-byte header holding the count, then cap ExchRec records, cap = f(count at the previous re-neighboring), derived alike by the
// sender (from what it sent) and the receiver (from what it got) — and every count (leavers, hole fillers, arrivals kept, the
// running nlocal) stays in device memory (`est`) until ONE read-back at the end of the three dimensions. est layout (ints):
//   [0] nlocal now  [1] overflow (max over the ranks)  [2] leavers of the current dimension  [3] 1 + first dimension that was NOT applied
//   [4+d] leavers of dimension d  [10+2d+dir] records received  [20] broken invariant (cannot happen by construction; fatal)
// Overflow protocol (a rank has more leavers than its message holds: more than 4x the previous migration + 4096 atoms): the sender
// sees that BEFORE anything is overwritten (k_ex_leavers only reads the atoms), the flag is max-reduced over the ranks on the stream
// before the first kernel that moves atoms (k_ex_fill), and from then on every rank skips k_ex_fill / k_ex_arrive of this and the
// following dimensions while the fixed-size messages keep flowing (matched sends and receives). The host then finds every rank in the
// same state — dimensions below est[3]-1 applied, the others untouched — and finishes them with the count-handshake path.
// ---------------------------------------------------------------------------------------------------
#define EST_NLOCAL 0
#define EST_OVF 1
#define EST_NSEND 2
#define EST_FAILD 3
#define EST_SEND_D 4
#define EST_RECV 10
#define EST_FATAL 20
#define EMSG_HEADER 64
static inline int exch_msg_cap(const mmd_handle* h, int prev) { return h->opt_exchange_cap > 0 ? h->opt_exchange_cap : 4 * prev + 4096; }
static inline size_t exch_msg_bytes(int cap) { return ((size_t)EMSG_HEADER + (size_t)cap * sizeof(ExchRec) + 63) & ~(size_t)63; }

__global__ __launch_bounds__(256) void k_ex_count(const real4* __restrict__ x, const int* __restrict__ est, int dim, real lo, real hi, int* __restrict__ cnt)
{
  __shared__ int lds[17];
  const int n = est[EST_NLOCAL];
  const int base = blockIdx.x * CP_TILE + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) if(base + k < n) { const real4 p = x[base + k]; const real v = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); c += (v < lo || v >= hi) ? 1 : 0; }
  int tot;
  block_incl_scan(c, lds, &tot);
  if(threadIdx.x == 0) cnt[blockIdx.x] = tot;
}
// ascending list of the leavers + Atom::pack_exchange (ref/atom.cpp:228-239) of each into the message
__global__ __launch_bounds__(256) void k_ex_leavers(const real4* __restrict__ x, const real* __restrict__ v, const int* __restrict__ tag,
                                                    int* __restrict__ est, int dim, real lo, real hi, const int* __restrict__ cnt,
                                                    int* __restrict__ leavers, unsigned char* __restrict__ msg, int cap, int d_index, int grid_atoms)
{
  __shared__ int lds[17];
  const int n = est[EST_NLOCAL];
  const int off = block_prefix_total(cnt, blockIdx.x, lds);
  const int base = blockIdx.x * CP_TILE + threadIdx.x * 4;
  bool fl[4];
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) {
    fl[k] = false;
    if(base + k < n) { const real4 p = x[base + k]; const real q = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); fl[k] = q < lo || q >= hi; }
    c += fl[k] ? 1 : 0;
  }
  int tot;
  const int inc = block_incl_scan(c, lds, &tot);
  int pos = off + inc - c;
  ExchRec* __restrict__ rec = (ExchRec*)(msg + EMSG_HEADER);
#pragma unroll
  for(int k = 0; k < 4; k++) {
    if(fl[k]) {
      if(pos < cap) {
        const int i = base + k;
        leavers[pos] = i;
        const real4 p = x[i];
        ExchRec r;
        r.x = p.x; r.y = p.y; r.z = p.z; r.w = p.w;
        r.vx = v[3 * (size_t)i + 0]; r.vy = v[3 * (size_t)i + 1]; r.vz = v[3 * (size_t)i + 2];
        r.tag = tag[i]; r.pad = 0;
        rec[pos] = r;
      }
      pos++;
    }
  }
  if(blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const int nsend = off + tot;
    est[EST_NSEND] = nsend; est[EST_SEND_D + d_index] = nsend;
    *(int*)msg = nsend;
    if(nsend > cap) est[EST_OVF] = 1;                         // more leavers than the message holds: nothing has been moved yet
    if(n > grid_atoms) est[EST_FATAL] = 1;                    // (more atoms than the launch covers: the host sized the grid from a bound)
  }
}
// the k-th hole (leaver below the new end) takes the k-th stayer of the tail: Atom::copy (ref/comm.cpp:491-509). One workgroup:
// the tail is as long as the leaver list (a few thousand atoms)
__global__ __launch_bounds__(1024) void k_ex_fill(real4* __restrict__ x, real* __restrict__ v, int* __restrict__ type, int* __restrict__ tag,
                                                  int* __restrict__ est, int dim, real lo, real hi, const int* __restrict__ leavers, int cap)
{
  __shared__ int lds[17];
  if(est[EST_OVF]) {                              // (uniform: some rank's message overflowed — this and the later dimensions are left to the handshake path)
    if(threadIdx.x == 0 && est[EST_FAILD] == 0) est[EST_FAILD] = dim + 1;
    return;
  }
  const int n = est[EST_NLOCAL], nsend = min(est[EST_NSEND], cap);
  const int t0 = n - nsend;
  int done = 0;                                   // stayers of the tail seen so far (uniform)
  for(int b = 0; b < nsend; b += 1024) {
    const int i = t0 + b + (int)threadIdx.x;
    bool stay = false;
    real4 p = real4{0, 0, 0, 0};
    if(b + (int)threadIdx.x < nsend) { p = x[i]; const real q = dim == 0 ? p.x : (dim == 1 ? p.y : p.z); stay = !(q < lo || q >= hi); }
    int tot;
    const int inc = block_incl_scan(stay ? 1 : 0, lds, &tot);
    if(stay) {
      const int dst = leavers[done + inc - 1];    // (ascending leavers: the holes below the new end are its first entries)
      x[dst] = p;
      v[3 * (size_t)dst + 0] = v[3 * (size_t)i + 0]; v[3 * (size_t)dst + 1] = v[3 * (size_t)i + 1]; v[3 * (size_t)dst + 2] = v[3 * (size_t)i + 2];
      type[dst] = type[i];
      tag[dst] = tag[i];
    }
    done += tot;
    __syncthreads();
  }
  if(threadIdx.x == 0) est[EST_NLOCAL] = t0;
}
// arrivals that fall inside my box in this dimension are appended in message order (ref/comm.cpp:566-571, Atom::unpack_exchange
// ref/atom.cpp:241-254); message 0 first, then message 1 (grids wider than 2 receive from both sides). One workgroup.
__global__ __launch_bounds__(1024) void k_ex_arrive(real4* __restrict__ x, real* __restrict__ v, int* __restrict__ type, int* __restrict__ tag,
                                                    int* __restrict__ est, int dim, real lo, real hi, const unsigned char* __restrict__ m0, int cap0,
                                                    const unsigned char* __restrict__ m1, int cap1, int cap_atoms, int d_index)
{
  __shared__ int lds[17];
  if(est[EST_OVF]) return;
  int n = est[EST_NLOCAL];
  for(int q = 0; q < 2; q++) {
    const unsigned char* __restrict__ m = q ? m1 : m0;
    if(m == nullptr) continue;
    const int cap = q ? cap1 : cap0, raw = *(const int*)m, cnt = min(max(raw, 0), cap);
    const ExchRec* __restrict__ rec = (const ExchRec*)(m + EMSG_HEADER);
    for(int b = 0; b < cnt; b += 1024) {
      const int k = b + (int)threadIdx.x;
      bool keep = false;
      ExchRec r;
      if(k < cnt) { r = rec[k]; const real c = dim == 0 ? r.x : (dim == 1 ? r.y : r.z); keep = c >= lo && c < hi; }
      int tot;
      const int inc = block_incl_scan(keep ? 1 : 0, lds, &tot);
      if(keep) {
        const int i = n + inc - 1;
        if(i < cap_atoms) {
          x[i] = real4{r.x, r.y, r.z, r.w};
          v[3 * (size_t)i + 0] = r.vx; v[3 * (size_t)i + 1] = r.vy; v[3 * (size_t)i + 2] = r.vz;
          type[i] = (int)r.w;
          tag[i] = r.tag;
        }
      }
      n += tot;
      __syncthreads();
    }
    // (raw > cap: the sender would have raised the overflow flag before this kernel ran; n > cap_atoms: the arrays were sized for every cap)
    if(threadIdx.x == 0) { est[EST_RECV + 2 * d_index + q] = raw; if(raw > cap) est[EST_FATAL] = 1; }
  }
  if(threadIdx.x == 0) { if(n > cap_atoms) est[EST_FATAL] = 1; est[EST_NLOCAL] = min(n, cap_atoms); }
}
__global__ void k_ex_init(int* __restrict__ est, int nlocal)
{
  if(threadIdx.x < 32) est[threadIdx.x] = threadIdx.x == EST_NLOCAL ? nlocal : 0;
}

// a flag word in device memory becomes its maximum over the ranks, on the stream (RCCL) / through the host (test transport)
static int reduce_flag_max(mmd_handle* h, int* dflag)
{
  if(h->rccl) { NCCL_TRY(ncclAllReduce(dflag, dflag, 1, ncclInt, ncclMax, (ncclComm_t)h->rccl, h->stream)); return 0; }
  HIP_TRY(hipMemcpyAsync(h->h_flags + 14, dflag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync_transport(h));
  double v = h->h_flags[14] ? 1.0 : 0.0;
  MMD_TRY(mmd_transport_allreduce(h, &v, 1));
  h->h_flags[14] = v > 0.0 ? 1 : 0;
  HIP_TRY(hipMemcpyAsync(dflag, h->h_flags + 14, sizeof(int), hipMemcpyHostToDevice, h->stream));
  return 0;
}

// returns 1 = done, 0 = the caller runs the handshake path from dimension *resume_dim on (0: path not applicable; d > 0: a
// message overflowed in dimension d, the dimensions below it are done), < 0 error
static int exchange_multi_fast(mmd_handle* h, int* resume_dim)
{
  *resume_dim = 0;
  if(h->nprocs == 1 || h->opt_safe_exchange || !h->ex_prev_valid || !(h->rccl || h->host_sr)) return 0;
  int cap_s[3] = {0, 0, 0}, cap_r[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  int arrivals_max = 0;
  for(int d = 0; d < 3; d++) {
    if(h->procgrid[d] == 1) continue;
    cap_s[d] = exch_msg_cap(h, h->ex_prev_send[d]);
    cap_r[d][0] = exch_msg_cap(h, h->ex_prev_recv[d][0]);
    cap_r[d][1] = h->procgrid[d] > 2 ? exch_msg_cap(h, h->ex_prev_recv[d][1]) : 0;
    arrivals_max += cap_r[d][0] + cap_r[d][1];
  }
  const int nl0 = h->nlocal;
  MMD_TRY(mmd_ensure_atoms(h, nl0 + arrivals_max + 1, true));
  MMD_TRY(h->est.ensure(64, false, h->stream));
  hipLaunchKernelGGL(k_ex_init, dim3(1), dim3(64), 0, h->stream, h->est.p, nl0);
  int bound = nl0;                                           // host-side bound of nlocal (grids, capacities)
  for(int d = 0; d < 3; d++) {
    if(h->procgrid[d] == 1) continue;
    const real lo = h->lo[d], hi = h->hi[d];
    const int nt = div_up(std::max(bound, 1), CP_TILE);
    const size_t bs = exch_msg_bytes(cap_s[d]), br0 = exch_msg_bytes(cap_r[d][0]), br1 = cap_r[d][1] ? exch_msg_bytes(cap_r[d][1]) : 0;
    MMD_TRY(h->flag_tmp.ensure((size_t)nt + 8, false, h->stream));
    MMD_TRY(h->ex_list.ensure((size_t)cap_s[d] + 8, false, h->stream));
    MMD_TRY(h->buf_send.ensure(bs / sizeof(real) + 16, false, h->stream));
    MMD_TRY(h->buf_recv.ensure((br0 + br1) / sizeof(real) + 16, false, h->stream));
    unsigned char* smsg = (unsigned char*)h->buf_send.p;
    unsigned char* rmsg0 = (unsigned char*)h->buf_recv.p;
    unsigned char* rmsg1 = br1 ? rmsg0 + br0 : nullptr;
    hipLaunchKernelGGL(k_ex_count, dim3(nt), dim3(256), 0, h->stream, h->x.p, h->est.p, d, lo, hi, h->flag_tmp.p);
    hipLaunchKernelGGL(k_ex_leavers, dim3(nt), dim3(256), 0, h->stream, h->x.p, h->v.p, h->tag.p, h->est.p, d, lo, hi, h->flag_tmp.p, h->ex_list.p, smsg,
                       cap_s[d], d, nt * CP_TILE);
    HIP_TRY(hipGetLastError());
    MMD_TRY(reduce_flag_max(h, h->est.p + EST_OVF));           // (before the first kernel of this dimension that moves atoms)
    hipLaunchKernelGGL(k_ex_fill, dim3(1), dim3(1024), 0, h->stream, h->x.p, h->v.p, h->type.p, h->tag.p, h->est.p, d, lo, hi, h->ex_list.p, cap_s[d]);
    HIP_TRY(hipGetLastError());
    if(br1) {
      const void* ds[2] = {smsg, smsg};
      void* dr[2] = {rmsg0, rmsg1};
      const size_t nsb[2] = {bs, bs}, nrb[2] = {br0, br1};
      const int dest[2] = {h->procneigh[d][0], h->procneigh[d][1]}, src[2] = {h->procneigh[d][1], h->procneigh[d][0]};
      MMD_TRY(mmd_transport_sendrecv_pair(h, ds, nsb, dest, dr, nrb, src));
    } else
      MMD_TRY(mmd_transport_sendrecv(h, smsg, bs, h->procneigh[d][0], rmsg0, br0, h->procneigh[d][1]));
    hipLaunchKernelGGL(k_ex_arrive, dim3(1), dim3(1024), 0, h->stream, h->x.p, h->v.p, h->type.p, h->tag.p, h->est.p, d, lo, hi,
                       (const unsigned char*)rmsg0, cap_r[d][0], (const unsigned char*)rmsg1, cap_r[d][1], h->nmax, d);
    HIP_TRY(hipGetLastError());
    bound += cap_r[d][0] + cap_r[d][1];
  }
  HIP_TRY(hipMemcpyAsync(h->h_flags_big, h->est.p, 32 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  const int* e = h->h_flags_big;
  if(e[EST_FATAL]) { mmd_set_error("Comm::exchange: the handshake-free path broke its own sizing (message or atom capacity)"); return -1; }
  const int d_end = e[EST_OVF] ? e[EST_FAILD] - 1 : 3;         // dimensions [0, d_end) were applied, on every rank alike
  if(e[EST_OVF] && (d_end < 0 || d_end > 2)) { mmd_set_error("Comm::exchange: overflow flag without a dimension"); return -1; }
  h->nlocal = e[EST_NLOCAL];
  for(int d = 0; d < d_end; d++) {
    if(h->procgrid[d] == 1) continue;
    h->ex_prev_send[d] = e[EST_SEND_D + d];
    h->ex_prev_recv[d][0] = e[EST_RECV + 2 * d];
    h->ex_prev_recv[d][1] = e[EST_RECV + 2 * d + 1];
  }
  if(e[EST_OVF]) { *resume_dim = d_end; h->ex_overflows++; return 0; }
  h->ex_fast++;
  return 1;
}

extern "C" int mmd_comm_exchange(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  MMD_TRY(mmd_atom_pbc(h));
  h->nghost = 0;                       // ghost slots are reused by arrivals; borders() rebuilds them next
  h->dh.ready = false; h->dh.pending = false;
  int d_first = 0;                     // (> 0: the handshake-free path stopped at an overflowing dimension; finish from there)
  {
    const int rc = exchange_multi_fast(h, &d_first);
    if(rc != 0) return rc < 0 ? rc : 0;
  }
  static_assert(sizeof(ExchRec) % sizeof(real) == 0, "ExchRec must be a whole number of reals");
  const size_t rec_reals = sizeof(ExchRec) / sizeof(real);
  DevArr<int> leavers, fillers, keep;
  for(int d = d_first; d < 3; d++) {
    if(h->procgrid[d] == 1) continue;
    const real lo = h->lo[d], hi = h->hi[d];
    const int nlocal = h->nlocal;
    int nsend = 0;
    MMD_TRY(compact(h, LeavePred{h->x.p, d, lo, hi}, 0, nlocal, leavers, &nsend));
    MMD_TRY(h->buf_send.ensure(rec_reals * nsend + 8, false, h->stream));
    if(nsend) {
      hipLaunchKernelGGL(k_pack_exchange, dim3(div_up(nsend, 256)), dim3(256), 0, h->stream, h->x.p, h->v.p, h->tag.p, leavers.p, nsend, (ExchRec*)h->buf_send.p);
      int nfill = 0;
      MMD_TRY(compact(h, StayPred{h->x.p, d, lo, hi}, nlocal - nsend, nsend, fillers, &nfill));
      // leavers are ascending, so the holes (index < nlocal-nsend) are exactly its first nfill entries
      if(nfill) hipLaunchKernelGGL(k_fill_holes, dim3(div_up(nfill, 256)), dim3(256), 0, h->stream, h->x.p, h->v.p, h->type.p, h->tag.p, leavers.p, fillers.p, nfill);
      HIP_TRY(hipGetLastError());
    }
    h->nlocal = nlocal - nsend;
    if(h->opt_safe_exchange) {
      // Comm::exchange_all (ref/comm.cpp:599-689, entered from :366-367 when do_safeexchange): the leavers of this dimension are
      // offered to every rank within need[d] sub-domains, nearest first, alternating -i / +i (sendproc_exc / recvproc_exc =
      // MPI_Cart_shift(cartesian, d, i), :176-180), while `ineed < procgrid[d] - 1` (:656); each receiver keeps what falls inside
      // its own [lo, hi) (:675-681). The buffer is packed once (above), arrivals are appended swap by swap.
      for(int ineed = 0; ineed < 2 * h->need[d]; ineed++) {
        if(!(ineed < h->procgrid[d] - 1)) continue;
        const int dist = ineed / 2 + 1;
        int lo_c[3] = {h->myloc[0], h->myloc[1], h->myloc[2]}, hi_c[3] = {h->myloc[0], h->myloc[1], h->myloc[2]};
        lo_c[d] -= dist; hi_c[d] += dist;
        const int below = cart_rank(h->procgrid, lo_c[0], lo_c[1], lo_c[2]), above = cart_rank(h->procgrid, hi_c[0], hi_c[1], hi_c[2]);
        const int dest = ineed % 2 == 0 ? below : above, src = ineed % 2 == 0 ? above : below;
        int nrecv = 0;
        MMD_TRY(mmd_transport_sendrecv_counts(h, nsend, dest, &nrecv, src));
        MMD_TRY(h->buf_recv.ensure(rec_reals * nrecv + 8, false, h->stream));
        MMD_TRY(mmd_transport_sendrecv(h, h->buf_send.p, (size_t)nsend * sizeof(ExchRec), dest, h->buf_recv.p, (size_t)nrecv * sizeof(ExchRec), src));
        int nkeep = 0;
        MMD_TRY(compact(h, ArrivePred{(const ExchRec*)h->buf_recv.p, d, lo, hi}, 0, nrecv, keep, &nkeep));
        if(nkeep) {
          MMD_TRY(mmd_ensure_atoms(h, h->nlocal + nkeep + 1, true));
          hipLaunchKernelGGL(k_unpack_exchange, dim3(div_up(nkeep, 256)), dim3(256), 0, h->stream, (const ExchRec*)h->buf_recv.p, keep.p, nkeep,
                             h->nlocal, h->x.p, h->v.p, h->type.p, h->tag.p);
          HIP_TRY(hipGetLastError());
          h->nlocal += nkeep;
        }
      }
      continue;
    }
    // send towards -1, receive from +1; and the other way round when the grid is wider than 2 (ref :521-543): both
    // directions share one count handshake (one host sync) and one payload group
    int nrecv1 = 0, nrecv2 = 0;
    if(h->procgrid[d] > 2) {
      const int ns[2] = {nsend, nsend}, dest[2] = {h->procneigh[d][0], h->procneigh[d][1]}, src[2] = {h->procneigh[d][1], h->procneigh[d][0]};
      int nr[2] = {0, 0};
      MMD_TRY(mmd_transport_sendrecv_counts_pair(h, ns, dest, nr, src));
      nrecv1 = nr[0]; nrecv2 = nr[1];
      MMD_TRY(h->buf_recv.ensure(rec_reals * (nrecv1 + nrecv2) + 8, false, h->stream));
      const void* dsend[2] = {h->buf_send.p, h->buf_send.p};
      void* drecv[2] = {h->buf_recv.p, (ExchRec*)h->buf_recv.p + nrecv1};
      const size_t bs[2] = {(size_t)nsend * sizeof(ExchRec), (size_t)nsend * sizeof(ExchRec)};
      const size_t br[2] = {(size_t)nrecv1 * sizeof(ExchRec), (size_t)nrecv2 * sizeof(ExchRec)};
      MMD_TRY(mmd_transport_sendrecv_pair(h, dsend, bs, dest, drecv, br, src));
    } else {
      MMD_TRY(mmd_transport_sendrecv_counts(h, nsend, h->procneigh[d][0], &nrecv1, h->procneigh[d][1]));
      MMD_TRY(h->buf_recv.ensure(rec_reals * nrecv1 + 8, false, h->stream));
      MMD_TRY(mmd_transport_sendrecv(h, h->buf_send.p, (size_t)nsend * sizeof(ExchRec), h->procneigh[d][0], h->buf_recv.p,
                                     (size_t)nrecv1 * sizeof(ExchRec), h->procneigh[d][1]));
    }
    const int nrecv = nrecv1 + nrecv2;
    h->ex_prev_send[d] = nsend; h->ex_prev_recv[d][0] = nrecv1; h->ex_prev_recv[d][1] = nrecv2;      // (size the next exchange's fixed messages)
    int nkeep = 0;
    MMD_TRY(compact(h, ArrivePred{(const ExchRec*)h->buf_recv.p, d, lo, hi}, 0, nrecv, keep, &nkeep));
    if(nkeep) {
      MMD_TRY(mmd_ensure_atoms(h, h->nlocal + nkeep + 1, true));
      hipLaunchKernelGGL(k_unpack_exchange, dim3(div_up(nkeep, 256)), dim3(256), 0, h->stream, (const ExchRec*)h->buf_recv.p, keep.p, nkeep,
                         h->nlocal, h->x.p, h->v.p, h->type.p, h->tag.p);
      HIP_TRY(hipGetLastError());
      h->nlocal += nkeep;
    }
  }
  if(h->nprocs > 1) HIP_TRY(mmd_stream_sync(h));     // (the scratch arrays below are only allocated when a dimension is split)
  leavers.release(); fillers.release(); keep.release();
  // the fixed-size messages of the next exchange are sized from THIS one's counts — but only from an exchange inside a run: the one of
  // the set-up (mmd_sim_initial) moves nobody, while at the first re-neighboring half a lattice plane may leave through every face
  // (FCC atoms are created exactly on the sub-domain faces: ~2 ny nz atoms, far beyond 4 x 0 + 4096 at -s 80 per rank)
  if(!h->opt_safe_exchange && h->in_reneighbor) h->ex_prev_valid = true;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Comm::borders (ref/comm.cpp:700-883) — ghost atoms + send lists, swap by swap (later swaps forward
// ghosts received by earlier ones, so the order x-,x+,y-,y+,z-,z+ is kept)
// ---------------------------------------------------------------------------------------------------
// image code of a ghost = (sx+2) + 5*(sy+2) + 25*(sz+2): accumulated periodic shifts in box lengths
__device__ __forceinline__ int image_add(int code, int px, int py, int pz)
{
  const int sx = code % 5 + px, sy = (code / 5) % 5 + py, sz = code / 25 + pz;
  return min(max(sx, 0), 4) + 5 * min(max(sy, 0), 4) + 25 * min(max(sz, 0), 4);
}
#define IMAGE_NONE 62      // (0,0,0)

// Atom::pack_border (ref/atom.cpp:197-214): {x+shift, type} -> dst (real4), image code -> dst_img
__global__ __launch_bounds__(256) void k_pack_border(const real4* __restrict__ x, const int* __restrict__ ghost_image,
                                                     const int* __restrict__ ghost_root, int nlocal,
                                                     const int* __restrict__ list, int n, real sx, real sy, real sz, int pbc_any,
                                                     int px, int py, int pz, real4* __restrict__ dst, int* __restrict__ dst_img,
                                                     int* __restrict__ dst_root, int* __restrict__ dst_type)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const int i = list[k];
  real4 p = x[i];
  if(pbc_any) { p.x += sx; p.y += sy; p.z += sz; }
  dst[k] = p;
  const int code = i < nlocal ? IMAGE_NONE : ghost_image[i - nlocal];
  dst_img[k] = image_add(code, px, py, pz);
  if(dst_root) dst_root[k] = i < nlocal ? i : ghost_root[i - nlocal];
  if(dst_type) dst_type[k] = (int)p.w;          // Atom::unpack_border tail (ref/atom.cpp:216-226) for a self swap
}

// One-rank fast path of Comm::communicate: every ghost is a periodic image of an owned atom; replay its chain of
// shifts (at most `need` per dimension, applied one box length at a time so the rounding equals the swap-by-swap
// result) in ONE kernel instead of 2*sum(need) dependent ones.
__global__ __launch_bounds__(256) void k_ghost_update(real4* __restrict__ x, int nlocal, int nghost, const int* __restrict__ root,
                                                      const int* __restrict__ image, real xprd, real yprd, real zprd)
{
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= nghost) return;
  real4 p = x[root[g]];
  const int code = image[g];
  const int sx = code % 5 - 2, sy = (code / 5) % 5 - 2, sz = code / 25 - 2;
  for(int q = 0; q < (sx < 0 ? -sx : sx); q++) p.x += sx < 0 ? -xprd : xprd;
  for(int q = 0; q < (sy < 0 ? -sy : sy); q++) p.y += sy < 0 ? -yprd : yprd;
  for(int q = 0; q < (sz < 0 ? -sz : sz); q++) p.z += sz < 0 ? -zprd : zprd;
  x[nlocal + g] = p;
}
// Atom::unpack_border tail (ref/atom.cpp:216-226): integer type array of the new ghosts
__global__ __launch_bounds__(256) void k_ghost_types(const real4* __restrict__ x, int first, int n, int* __restrict__ type)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  type[first + k] = (int)x[first + k].w;
}

// ---------------------------------------------------------------------------------------------------
// Device-resident Comm::borders (six swaps, need = 1 in every dimension): nothing but the final counts has to reach the host.
// The swaps run as count / scatter kernel pairs whose ranges, offsets and running ghost count live in device memory (`bst`),
// sized by the previous re-neighboring's counts (+50 %); ONE synchronisation at the end (or none: the neighbor build's own
// read-back brings bst along) returns the counts and an overflow flag (then the swap-by-swap path below redoes the work with grown
// arrays). A periodic self swap writes its ghosts in place, a swap with another rank goes through a fixed-size message (below).
// Same selections in the same order as the swap-by-swap path => identical send lists and ghosts. bst layout (ints):
//   [0] nb = owned atoms inside any send slab   [1] overflow flag   [4+s] sendnum of swap s   [12+s] recvnum   [30+s] ghosts before swap s
// ---------------------------------------------------------------------------------------------------
#define BST_NB 0
#define BST_OVF 1
#define BST_SEND 4
#define BST_RECV 12
#define BST_GHOSTS 30
// Several ranks: a dimension whose partners are other ranks runs the same count / scatter pair, but the selection goes into one
// MESSAGE per swap — a 64-byte header holding the count, then cap {x+shift, type} records, then cap image codes — of a FIXED size
// cap = f(count of this swap at the previous re-neighboring) that both sides derive alike (the sender from its previous sendnum,
// the receiver from its previous recvnum: the same number), so no count handshake and no host synchronisation is needed to post
// the receive; k_border_unpack appends min(count, cap) ghosts behind the running ghost count in bst and records recvnum
// ([BST_RECV+s]). A count beyond its cap raises the overflow flag, which is max-reduced over the ranks before anybody reads it:
// then every rank redoes the borders swap by swap (borders_general).
#define BMSG_HEADER 64
static inline int border_msg_cap(int prev) { return prev + prev / 16 + 2048; }
static inline size_t border_msg_bytes(int cap) { return ((size_t)BMSG_HEADER + (size_t)cap * (sizeof(real4) + sizeof(int)) + 63) & ~(size_t)63; }
struct SlabSet { real lo[6], hi[6]; int dim[6]; int n; };

__device__ __forceinline__ bool in_any_slab(const real4 p, const SlabSet& S)
{
  bool in = false;
  for(int s = 0; s < S.n; s++) { const real c = S.dim[s] == 0 ? p.x : (S.dim[s] == 1 ? p.y : p.z); in = in || (c >= S.lo[s] && c <= S.hi[s]); }
  return in;
}

// (both kernels read x the way it lies in memory: wavefront w of a workgroup takes atoms [256 w, 256 w + 256) of the workgroup's 1024 in four
//  rounds of 64 consecutive atoms — whole lines per load instruction — and orders its hits with ballots; four atoms per thread at a
//  stride of 128 bytes between the lanes ran at 2.2 TB/s)
__global__ __launch_bounds__(256) void k_bnd_count(const real4* __restrict__ x, int nlocal, SlabSet S, int* __restrict__ cnt, int* __restrict__ bst)
{
  __shared__ int s_w[4];
  if(blockIdx.x == 0 && threadIdx.x < 64) bst[threadIdx.x] = 0;          // the state words of this borders pass (first written by k_bnd_scatter)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  int c = 0;
#pragma unroll
  for(int r = 0; r < 4; r++) {
    const int i = base + r * 64;
    const bool in = i < nlocal && in_any_slab(x[i < nlocal ? i : 0], S);
    c += __popcll(__builtin_amdgcn_ballot_w64(in));
  }
  if(lane == 0) s_w[wave] = c;
  __syncthreads();
  if(threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_bnd_scatter(const real4* __restrict__ x, int nlocal, SlabSet S, const int* __restrict__ cnt,
                                                     int* __restrict__ bnd, int* __restrict__ bst, int est_nb)
{
  __shared__ int lds[17];
  __shared__ int s_w[4];
  const int off = block_prefix_total(cnt, blockIdx.x, lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  unsigned long long m[4];
  int c = 0;
#pragma unroll
  for(int r = 0; r < 4; r++) {
    const int i = base + r * 64;
    const bool in = i < nlocal && in_any_slab(x[i < nlocal ? i : 0], S);
    m[r] = __builtin_amdgcn_ballot_w64(in);
    c += __popcll(m[r]);
  }
  if(lane == 0) s_w[wave] = c;
  __syncthreads();
  int pos = off;
  for(int w = 0; w < wave; w++) pos += s_w[w];
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for(int r = 0; r < 4; r++) {
    if((m[r] >> lane) & 1ull) bnd[pos + __popcll(m[r] & below)] = base + r * 64;
    pos += __popcll(m[r]);
  }
  if(blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    bst[BST_NB] = off + tot;
    if(off + tot > est_nb) bst[BST_OVF] = 1;        // (the swap kernels' grids were sized for est_nb boundary atoms)
  }
}
// The two swaps of a dimension (sw0 towards -1, sw0+1 towards +1) select from the same atoms — owned boundary atoms and the
// ghosts of the EARLIER dimensions — so one launch serves both (blockIdx.y = which of the two).
struct SwapPair { real lo[2], hi[2], sx[2], sy[2], sz[2]; int pbc_any[2], px[2], py[2], pz[2]; int cap_list[2]; int* sendlist[2];
                  unsigned char* msg[2]; int cap_msg[2]; };      // (remote partners: the outgoing messages and their record capacity)

__global__ __launch_bounds__(256) void k_swap_count(const real4* __restrict__ x, const int* __restrict__ bnd, const int* __restrict__ bst,
                                                    int nlocal, int sw0, int dim, SwapPair P, int* __restrict__ cnt, int ncnt)
{
  __shared__ int lds[17];
  const int y = blockIdx.y;
  const real lo = P.lo[y], hi = P.hi[y];
  const int nb = bst[BST_NB], n = nb + bst[BST_GHOSTS + sw0];
  // (wavefront w takes candidates [256 w, 256 w + 256) of the workgroup's 1024 in four rounds of 64 consecutive ones, see k_bnd_count)
  __shared__ int s_w[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  int c = 0;
#pragma unroll
  for(int r = 0; r < 4; r++) {
    const int q = base + r * 64;
    bool in = false;
    if(q < n) {
      const real4 p = x[q < nb ? bnd[q] : nlocal + (q - nb)];
      const real v = dim == 0 ? p.x : (dim == 1 ? p.y : p.z);
      in = v >= lo && v <= hi;
    }
    c += __popcll(__builtin_amdgcn_ballot_w64(in));
  }
  if(lane == 0) s_w[wave] = c;
  __syncthreads();
  if(threadIdx.x == 0) cnt[y * ncnt + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  (void)lds;
}
template <bool REMOTE>
__global__ __launch_bounds__(256) void k_swap_scatter(real4* __restrict__ x, const int* __restrict__ bnd, int* __restrict__ bst, int nlocal,
                                                      int sw0, int dim, SwapPair P, const int* __restrict__ cnt, int ncnt,
                                                      int cap_atoms, int cap_ghost, int* __restrict__ ghost_image, int* __restrict__ ghost_root,
                                                      int* __restrict__ type)
{
  __shared__ int lds[17];
  const int y = blockIdx.y;
  const real lo = P.lo[y], hi = P.hi[y];
  const int nb = bst[BST_NB], n = nb + bst[BST_GHOSTS + sw0];
  const bool last = blockIdx.x == gridDim.x - 1;
  if((long long)blockIdx.x * CP_TILE >= n && !last) return;
  // ghosts in front of this swap's: those of the earlier dimensions, plus ALL of the pair's first swap for the second
  int nghost = bst[BST_GHOSTS + sw0];
  if(!REMOTE && y == 1) nghost += block_prefix_total(cnt, ncnt, lds);
  const int nall = nlocal + nghost;
  real4* __restrict__ m_rec = REMOTE ? (real4*)(P.msg[y] + BMSG_HEADER) : nullptr;
  int* __restrict__ m_img = REMOTE ? (int*)(P.msg[y] + BMSG_HEADER + (size_t)P.cap_msg[y] * sizeof(real4)) : nullptr;
  const int off = block_prefix_total(cnt + y * ncnt, blockIdx.x, lds);
  __shared__ int s_w[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  bool fl[4];
  int idx[4];
  real4 pp[4];
  unsigned long long mk[4];
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) {
    const int q = base + k * 64;
    fl[k] = false; idx[k] = 0;
    if(q < n) {
      idx[k] = q < nb ? bnd[q] : nlocal + (q - nb);
      pp[k] = x[idx[k]];
      const real v = dim == 0 ? pp[k].x : (dim == 1 ? pp[k].y : pp[k].z);
      fl[k] = v >= lo && v <= hi;
    }
    mk[k] = __builtin_amdgcn_ballot_w64(fl[k]);
    c += __popcll(mk[k]);
  }
  if(lane == 0) s_w[wave] = c;
  __syncthreads();
  const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  int wpos = off;
  for(int w = 0; w < wave; w++) wpos += s_w[w];
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  bool ovf = false;
  int* __restrict__ sendlist = P.sendlist[y];
#pragma unroll
  for(int k = 0; k < 4; k++) {
    const int pos = wpos + __popcll(mk[k] & below);
    wpos += __popcll(mk[k]);
    if(fl[k]) {
      if(REMOTE) {                              // Atom::pack_border (ref/atom.cpp:197-214) into the message of this swap
        if(pos < P.cap_list[y] && pos < P.cap_msg[y]) {
          const int i = idx[k];
          real4 p = pp[k];
          if(P.pbc_any[y]) { p.x += P.sx[y]; p.y += P.sy[y]; p.z += P.sz[y]; }
          sendlist[pos] = i;
          m_rec[pos] = p;
          m_img[pos] = image_add(i < nlocal ? IMAGE_NONE : ghost_image[i - nlocal], P.px[y], P.py[y], P.pz[y]);
        } else ovf = true;
      } else
      if(pos < P.cap_list[y] && nall + pos < cap_atoms && nghost + pos < cap_ghost) {
        const int i = idx[k];
        real4 p = pp[k];
        if(P.pbc_any[y]) { p.x += P.sx[y]; p.y += P.sy[y]; p.z += P.sz[y]; }
        sendlist[pos] = i;
        x[nall + pos] = p;
        const int code = i < nlocal ? IMAGE_NONE : ghost_image[i - nlocal];
        ghost_image[nghost + pos] = image_add(code, P.px[y], P.py[y], P.pz[y]);
        ghost_root[nghost + pos] = i < nlocal ? i : ghost_root[i - nlocal];
        type[nall + pos] = (int)p.w;
      } else ovf = true;
    }
  }
  if(ovf) bst[BST_OVF] = 1;
  if(last) {                                  // (its prefix covers every other tile: off + tot is the swap's total)
    if((long long)gridDim.x * CP_TILE < n) bst[BST_OVF] = 1;        // the launch was sized for fewer candidates than there are
    if(threadIdx.x == 0) {
      bst[BST_SEND + sw0 + y] = off + tot;
      if(REMOTE) *(int*)P.msg[y] = off + tot;                 // header of the message: how many records follow
      else { bst[BST_RECV + sw0 + y] = off + tot; if(y == 1) bst[BST_GHOSTS + sw0 + 2] = nghost + off + tot; }
    }
  }
}

// Atom::unpack_border (ref/atom.cpp:216-226) for the two messages a rank receives in one dimension: ghosts of swap sw0 first, those
// of swap sw0+1 behind them (the order the swap-by-swap path appends them in)
__global__ __launch_bounds__(256) void k_border_unpack(real4* __restrict__ x, int* __restrict__ bst, int nlocal, int sw0,
                                                       const unsigned char* __restrict__ rmsg0, const unsigned char* __restrict__ rmsg1,
                                                       int cap0, int cap1, int cap_atoms, int cap_ghost, int* __restrict__ ghost_image,
                                                       int* __restrict__ type)
{
  const int y = blockIdx.y;
  const int raw0 = *(const int*)rmsg0, raw1 = *(const int*)rmsg1;
  const int c0 = min(max(raw0, 0), cap0), c1 = min(max(raw1, 0), cap1);
  const int base = bst[BST_GHOSTS + sw0] + (y ? c0 : 0), cnt = y ? c1 : c0, cap = y ? cap1 : cap0;
  const unsigned char* __restrict__ m = y ? rmsg1 : rmsg0;
  const real4* __restrict__ rec = (const real4*)(m + BMSG_HEADER);
  const int* __restrict__ img = (const int*)(m + BMSG_HEADER + (size_t)cap * sizeof(real4));
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k < cnt) {
    if(nlocal + base + k < cap_atoms && base + k < cap_ghost) {
      const real4 p = rec[k];
      x[nlocal + base + k] = p;
      ghost_image[base + k] = img[k];
      type[nlocal + base + k] = (int)p.w;
    } else bst[BST_OVF] = 1;
  }
  if(blockIdx.x == 0 && threadIdx.x == 0) {
    if((y ? raw1 : raw0) > cap) bst[BST_OVF] = 1;             // the sender selected more atoms than the message holds
    bst[BST_RECV + sw0 + y] = cnt;
    if(y == 1) bst[BST_GHOSTS + sw0 + 2] = bst[BST_GHOSTS + sw0] + c0 + c1;
  }
}

// ---------------------------------------------------------------------------------------------------
// One rank, every swap a periodic self swap (round 4): the six swaps in THREE launches instead of eight.
// A ghost is a periodic image of an owned atom ("root"), and whether swap q selects an atom or one of its earlier images depends on the
// root's own coordinate alone: the images made by the x swaps carry the root's y and z, those made by the y swaps its z (pack_border adds a
// shift of zero in the other dimensions). So the ghosts fall into 26 lists, one per non-empty combination (x swap | none, y swap | none,
// z swap | none), each holding the roots inside ALL its slabs in ascending index order, and the swap-by-swap sequence of ref/comm.cpp:364-597
// (a swap scans the owned atoms, then the ghosts of the earlier dimensions in index order) is these lists laid end to end in the order of
// BrdList::M below. k_brd_count: per workgroup of 1024 atoms the number of members of every list (+ row 26: atoms inside any slab) and a
// byte of slab bits per atom; k_brd_scan: exclusive scan of each row over the workgroups; k_brd_scatter: every member writes its ghost
// (position + the shifts of its swaps, image code, root, type) and the entry of the send list of the swap that made it — the index of the atom or
// earlier image it was copied from. Same ghosts, same send lists, same counts as the swap-by-swap path.
// ---------------------------------------------------------------------------------------------------
#define BRD_NL 26
#define BRD_ROWS 27
struct SelfSwaps { real lo[6], hi[6], sx[6], sy[6], sz[6]; int pbc_any[6], px[6], py[6], pz[6]; int cap_list[6]; int* sendlist[6]; };
// slab bits a list asks for (bit q = swap q), in ghost order: swap 0, swap 1, swap 2 over {owned, images of 0, of 1}, swap 3 likewise,
// swap 4 over {owned, images of 0, 1, 2 (three lists), 3 (three lists)}, swap 5 likewise
constexpr int brd_mask_of(int l)
{
  constexpr int M[BRD_NL] = {0x01, 0x02, 0x04, 0x05, 0x06, 0x08, 0x09, 0x0a, 0x10, 0x11, 0x12, 0x14, 0x15, 0x16, 0x18, 0x19, 0x1a,
                             0x20, 0x21, 0x22, 0x24, 0x25, 0x26, 0x28, 0x29, 0x2a};
  return M[l];
}
constexpr int brd_find(int m) { for(int l = 0; l < BRD_NL; l++) if(brd_mask_of(l) == m) return l; return -1; }
template <int L> struct BrdList {
  static constexpr int index = L;
  static constexpr int mask = brd_mask_of(L);
  static constexpr int swap = L < 1 ? 0 : L < 2 ? 1 : L < 5 ? 2 : L < 8 ? 3 : L < 17 ? 4 : 5;              // the swap that makes the list's ghosts
  // the list its ghosts are copied from (-1: from the owned atoms): the one whose mask is this mask without the bit of the making swap
  static constexpr int source = (mask & ~(1 << swap)) == 0 ? -1 : brd_find(mask & ~(1 << swap));
  static_assert(source < L, "a list is copied from an earlier one");
};
// f(BrdList<0>{}), ..., f(BrdList<25>{})
template <int L = 0, class Fn>
__device__ __forceinline__ void brd_for_lists(Fn&& f)
{
  if constexpr(L < BRD_NL) { f(BrdList<L>{}); brd_for_lists<L + 1>(f); }
}

__device__ __forceinline__ int brd_slab_bits(const real4 p, const SelfSwaps& W)
{
  int F = 0;
#pragma unroll
  for(int q = 0; q < 6; q++) {
    const real c = q < 2 ? p.x : (q < 4 ? p.y : p.z);
    F |= (c >= W.lo[q] && c <= W.hi[q]) ? (1 << q) : 0;          // ref/comm.cpp:411-413
  }
  return F;
}

__global__ __launch_bounds__(256) void k_brd_count(const real4* __restrict__ x, int nlocal, SelfSwaps W, unsigned char* __restrict__ bits,
                                                   int* __restrict__ cnt, int nblk, int* __restrict__ bst)
{
  __shared__ int s_c[4][BRD_ROWS];
  if(blockIdx.x == 0 && threadIdx.x < 64) bst[threadIdx.x] = 0;          // the state words of this borders pass (filled in by k_brd_scatter)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  int c[BRD_ROWS];
#pragma unroll
  for(int l = 0; l < BRD_ROWS; l++) c[l] = 0;
#pragma unroll
  for(int r = 0; r < 4; r++) {
    const int i = base + r * 64;
    const int F = i < nlocal ? brd_slab_bits(x[i], W) : 0;
    if(i < nlocal) bits[i] = (unsigned char)F;
    const unsigned long long any = __builtin_amdgcn_ballot_w64(F != 0);
    if(any == 0ull) continue;                                            // (interior atoms: nothing to count)
    brd_for_lists([&](auto L) {
      constexpr int l = decltype(L)::index, m = decltype(L)::mask;
      c[l] += __popcll(__builtin_amdgcn_ballot_w64((F & m) == m));
    });
    c[BRD_NL] += __popcll(any);
  }
  if(lane == 0) {
#pragma unroll
    for(int l = 0; l < BRD_ROWS; l++) s_c[wave][l] = c[l];
  }
  __syncthreads();
  if(threadIdx.x < BRD_ROWS) cnt[threadIdx.x * nblk + blockIdx.x] = s_c[0][threadIdx.x] + s_c[1][threadIdx.x] + s_c[2][threadIdx.x] + s_c[3][threadIdx.x];
}

// one workgroup per row: exclusive scan over the nblk workgroups of k_brd_count, the row's total behind it (tot[row])
__global__ __launch_bounds__(256) void k_brd_scan(int* __restrict__ cnt, int nblk, int* __restrict__ tot)
{
  __shared__ int lds[17];
  int* __restrict__ row = cnt + (size_t)blockIdx.x * nblk;
  int carry = 0;
  for(int c0 = 0; c0 < nblk; c0 += 1024) {
    const int b = c0 + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for(int k = 0; k < 4; k++) v[k] = b + k < nblk ? row[b + k] : 0;
    int total;
    const int inc = block_incl_scan(v[0] + v[1] + v[2] + v[3], lds, &total);
    int e = carry + inc - (v[0] + v[1] + v[2] + v[3]);
#pragma unroll
    for(int k = 0; k < 4; k++) { if(b + k < nblk) row[b + k] = e; e += v[k]; }
    carry += total;
  }
  if(threadIdx.x == 0) tot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_brd_scatter(real4* __restrict__ x, int nlocal, SelfSwaps W, const unsigned char* __restrict__ bits,
                                                     const int* __restrict__ cnt, int nblk, const int* __restrict__ tot, int* __restrict__ bst,
                                                     int cap_atoms, int cap_ghost, int* __restrict__ ghost_image, int* __restrict__ ghost_root,
                                                     int* __restrict__ type)
{
  __shared__ int s_c[4][BRD_NL];       // members of list l among wavefront w's 256 atoms
  __shared__ int s_first[BRD_NL];      // ghost number of the first image this WORKGROUP adds to list l
  __shared__ int s_sw[16];             // [q] first ghost of swap q (q = 6: all ghosts), [8 + q] ghosts of swap q
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  // every load whose address is known at the start goes out first: my atoms' slab bits; lane l < 27 of wavefront 0: total and scanned count of row l
  int F[4];
#pragma unroll
  for(int r = 0; r < 4; r++) F[r] = base + r * 64 < nlocal ? (int)bits[base + r * 64] : 0;
  int my_tot = 0, my_cnt = 0, next_any = 0;
  if(wave == 0 && lane < BRD_ROWS) {
    my_tot = tot[lane];
    my_cnt = cnt[lane * nblk + blockIdx.x];
    if(lane == BRD_NL) next_any = (int)blockIdx.x + 1 < nblk ? cnt[BRD_NL * nblk + blockIdx.x + 1] : my_tot;
  }
  const int Fany = F[0] | F[1] | F[2] | F[3];
  const unsigned U = wave_or_u((unsigned)Fany);                     // slab bits that occur among this wavefront's atoms
  // second round trip (issued before anything waits): the positions of my boundary atoms
  real4 p0[4];
#pragma unroll
  for(int r = 0; r < 4; r++) p0[r] = F[r] != 0 ? x[base + r * 64] : real4{0, 0, 0, 0};
  // ---- members of every list per wavefront
  brd_for_lists([&](auto L) {
    constexpr int l = decltype(L)::index, m = decltype(L)::mask;
    int c = 0;
    if((U & m) == m) {
#pragma unroll
      for(int r = 0; r < 4; r++) c += __popcll(__builtin_amdgcn_ballot_w64((F[r] & m) == m));
    }
    if(lane == 0) s_c[wave][l] = c;
  });
  // ---- where every list starts among the ghosts, where every swap's ghosts start: wavefront 0 from the 27 totals (lists laid end to end)
  bool ovf = false, empty = false;
  if(wave == 0) {
    const int lt = lane < BRD_NL ? my_tot : 0;
    const int lstart = wave_incl_scan(lt) - lt;                     // first ghost of list `lane`
    int sw_num[6], sw_first[7];
#pragma unroll
    for(int q = 0; q < 6; q++) sw_num[q] = 0;
    brd_for_lists([&](auto L) { sw_num[decltype(L)::swap] += __builtin_amdgcn_readlane(lt, decltype(L)::index); });
    sw_first[0] = 0;
#pragma unroll
    for(int q = 0; q < 6; q++) sw_first[q + 1] = sw_first[q] + sw_num[q];
    const int nghost = sw_first[6];
    ovf = nghost > cap_ghost || nlocal + nghost > cap_atoms;
#pragma unroll
    for(int q = 0; q < 6; q++) ovf = ovf || sw_num[q] > W.cap_list[q];
    if(lane < BRD_NL) s_first[lane] = lstart + my_cnt;
    if(lane == 0) {
#pragma unroll
      for(int q = 0; q < 7; q++) s_sw[q] = sw_first[q];
      s_sw[7] = ovf ? 1 : 0;
    }
    const int any0 = __builtin_amdgcn_readlane(my_cnt, BRD_NL), any1 = __builtin_amdgcn_readlane(next_any, BRD_NL);
    if(lane == 0) s_sw[15] = any0 == any1 ? 1 : 0;                  // no boundary atom in this workgroup
    if(blockIdx.x == 0 && lane == 0) {
      bst[BST_NB] = __builtin_amdgcn_readlane(my_tot, BRD_NL);
      bst[BST_OVF] = ovf ? 1 : 0;
#pragma unroll
      for(int q = 0; q < 6; q++) { bst[BST_SEND + q] = sw_num[q]; bst[BST_RECV + q] = sw_num[q]; bst[BST_GHOSTS + q] = sw_first[q]; }
      bst[BST_GHOSTS + 6] = nghost;
    }
  }
  __syncthreads();
  ovf = s_sw[7] != 0; empty = s_sw[15] != 0;
  if(ovf || empty || U == 0u) return;       // (arrays sized too small: the swap-by-swap path redoes the borders with grown ones) / nothing to add
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // ghost number of the first image THIS WAVEFRONT adds to list l: lane l works it out for all 26 lists at once (one pass over the LDS counts);
  // the list loop below fetches its two numbers with v_readlane
  int myoff = 0;
  if(lane < BRD_NL) {
    myoff = s_first[lane];
    for(int w = 0; w < wave; w++) myoff += s_c[w][lane];
  }
  brd_for_lists([&](auto L) {
    using LL = decltype(L);
    constexpr int l = LL::index, m = LL::mask, q = LL::swap, src = LL::source;
    if((U & m) != m) return;
    int o = __builtin_amdgcn_readlane(myoff, l);
    // (the image a member is copied from sits in list `src`: its ghost number is recounted here rather than kept from that list's turn)
    constexpr int ms = src < 0 ? 0 : brd_mask_of(src < 0 ? 0 : src);
    int os = src < 0 ? 0 : __builtin_amdgcn_readlane(myoff, src < 0 ? 0 : src);
    const int swf = s_sw[q];
#pragma unroll
    for(int r = 0; r < 4; r++) {
      const bool in = (F[r] & m) == m;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(in);
      const int g = o + __popcll(mm & below);
      o += __popcll(mm);
      int gs = 0;
      if(src >= 0) {
        const unsigned long long ss = __builtin_amdgcn_ballot_w64((F[r] & ms) == ms);
        gs = os + __popcll(ss & below);
        os += __popcll(ss);
      }
      if(in) {
        const int i = base + r * 64;
        // the chain of swaps this image went through, in order: position and image code exactly as pack_border accumulates them
        real4 p = p0[r];
        int code = IMAGE_NONE;
#pragma unroll
        for(int s = 0; s < 6; s++) {
          if(m & (1 << s)) {
            if(W.pbc_any[s]) { p.x += W.sx[s]; p.y += W.sy[s]; p.z += W.sz[s]; }
            code = image_add(code, W.px[s], W.py[s], W.pz[s]);
          }
        }
        x[nlocal + g] = p;
        ghost_image[g] = code;
        ghost_root[g] = i;
        type[nlocal + g] = (int)p.w;
        W.sendlist[q][g - swf] = src < 0 ? i : nlocal + gs;
      }
    }
  });
}

// ---------------------------------------------------------------------------------------------------
// Direct halo (DirectHalo, mmd_internal.hpp): Comm::communicate of a step on several ranks as ONE exchange.
// ---------------------------------------------------------------------------------------------------
// the 26 send lists of this rank, end to end in list order: owned atoms inside all slabs of the list, ascending (k_brd_count / k_brd_scan went before)
__global__ __launch_bounds__(256) void k_dh_lists(int nlocal, const unsigned char* __restrict__ bits, const int* __restrict__ cnt, int nblk,
                                                  const int* __restrict__ tot, int* __restrict__ idx, int* __restrict__ counts, int cap)
{
  __shared__ int s_c[4][BRD_NL];
  __shared__ int s_first[BRD_NL];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = blockIdx.x * CP_TILE + wave * 256 + lane;
  int F[4];
#pragma unroll
  for(int r = 0; r < 4; r++) F[r] = base + r * 64 < nlocal ? (int)bits[base + r * 64] : 0;
  int my_tot = 0, my_cnt = 0;
  if(wave == 0 && lane < BRD_NL) { my_tot = tot[lane]; my_cnt = cnt[lane * nblk + blockIdx.x]; }
  const unsigned U = wave_or_u((unsigned)(F[0] | F[1] | F[2] | F[3]));
  brd_for_lists([&](auto L) {
    constexpr int l = decltype(L)::index, m = decltype(L)::mask;
    int c = 0;
    if((U & m) == m) {
#pragma unroll
      for(int r = 0; r < 4; r++) c += __popcll(__builtin_amdgcn_ballot_w64((F[r] & m) == m));
    }
    if(lane == 0) s_c[wave][l] = c;
  });
  if(wave == 0) {
    const int lt = lane < BRD_NL ? my_tot : 0;
    const int lstart = wave_incl_scan(lt) - lt;
    if(lane < BRD_NL) s_first[lane] = lstart + my_cnt;
    if(blockIdx.x == 0 && lane < BRD_NL) counts[lane] = my_tot;
  }
  __syncthreads();
  if(U == 0u) return;
  int myoff = 0;
  if(lane < BRD_NL) {
    myoff = s_first[lane];
    for(int w = 0; w < wave; w++) myoff += s_c[w][lane];
  }
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  brd_for_lists([&](auto L) {
    constexpr int l = decltype(L)::index, m = decltype(L)::mask;
    if((U & m) != m) return;
    int o = __builtin_amdgcn_readlane(myoff, l);
#pragma unroll
    for(int r = 0; r < 4; r++) {
      const bool in = (F[r] & m) == m;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(in);
      if(in && o + __popcll(mm & below) < cap) idx[o + __popcll(mm & below)] = base + r * 64;          // (the host checks the total against cap when the lengths arrive)
      o += __popcll(mm);
    }
  });
}

struct DhMeta { int soff[27]; int self_base[26]; int sdst[26]; real sx[26], sy[26], sz[26]; };       // self_base >= 0: the list's target is this rank itself
// what = 0: positions {x + shift, type} of the listed atoms into the send buffer (the lists of one partner end to end); a list whose target is this
// rank goes straight into its ghost slots
__global__ __launch_bounds__(256) void k_dh_pack_x(real4* __restrict__ x, const int* __restrict__ idx, int total, DhMeta M, real4* __restrict__ buf, int nlocal)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= total) return;
  int l = 0;
#pragma unroll
  for(int q = 1; q < 26; q++) l += e >= M.soff[q] ? 1 : 0;
  real4 p = x[idx[e]];
  p.x += M.sx[l]; p.y += M.sy[l]; p.z += M.sz[l];
  if(M.self_base[l] >= 0) x[nlocal + M.self_base[l] + (e - M.soff[l])] = p; else buf[M.sdst[l] + (e - M.soff[l])] = p;
}
__global__ __launch_bounds__(256) void k_dh_pack_f(real* __restrict__ fp, const int* __restrict__ idx, int total, DhMeta M, real* __restrict__ buf, int nlocal)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= total) return;
  int l = 0;
#pragma unroll
  for(int q = 1; q < 26; q++) l += e >= M.soff[q] ? 1 : 0;
  const real v = fp[idx[e]];
  if(M.self_base[l] >= 0) fp[nlocal + M.self_base[l] + (e - M.soff[l])] = v; else buf[M.sdst[l] + (e - M.soff[l])] = v;
}
// the received lists (one message per partner) into the ghost slots: ghost g belongs to list l (rbase[l] <= g < rbase[l+1]), entry g - rbase[l] of it
struct DhUnpack { int rbase[27]; int rsrc[26]; };
template <typename T>
__global__ __launch_bounds__(256) void k_dh_unpack(T* __restrict__ dst, const T* __restrict__ rbuf, int nghost, DhUnpack U)
{
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= nghost) return;
  int l = 0;
#pragma unroll
  for(int q = 1; q < 26; q++) l += g >= U.rbase[q] ? 1 : 0;
  if(U.rsrc[l] >= 0) dst[g] = rbuf[U.rsrc[l] + (g - U.rbase[l])];
}

static bool dh_applies(const mmd_handle* h)
{
  if(!h->dh.opt || h->swaps.size() != 6 || !(h->nprocs > 1 || h->opt_force_transport) || !(h->rccl || h->host_sr)) return false;
  for(int q = 0; q < 6; q++) if(h->swaps[q].dim != q / 2) return false;
  return true;
}

static void dh_self_swaps(const mmd_handle* h, SelfSwaps& W)
{
  for(int q = 0; q < 6; q++) {
    const Swap& sw = h->swaps[q];
    W.lo[q] = sw.slablo; W.hi[q] = sw.slabhi;
    W.sx[q] = sw.pbc[0] * h->prd[0]; W.sy[q] = sw.pbc[1] * h->prd[1]; W.sz[q] = sw.pbc[2] * h->prd[2];
    W.pbc_any[q] = sw.pbc_any; W.px[q] = sw.pbc[0]; W.py[q] = sw.pbc[1]; W.pz[q] = sw.pbc[2];
    W.cap_list[q] = 0; W.sendlist[q] = nullptr;
  }
}
static int image_add_host(int code, int px, int py, int pz)
{
  const int sx = code % 5 + px, sy = (code / 5) % 5 + py, sz = code / 25 + pz;
  return std::min(std::max(sx, 0), 4) + 5 * std::min(std::max(sy, 0), 4) + 25 * std::min(std::max(sz, 0), 4);
}
// where every list goes and comes from: a list travels one step along each of its swaps' send directions (swap 2d towards -1, 2d+1 towards +1); the
// distinct partners in list order — sender and receiver enumerate them alike (target_l(A) = B <=> source_l(B) = A); shift and image code of its ghosts
static void dh_topology(mmd_handle* h)
{
  DirectHalo& D = h->dh;
  const bool forced = h->opt_force_transport != 0;
  for(int l = 0; l < 26; l++) {
    int off[3] = {0, 0, 0};
    D.shift[l][0] = D.shift[l][1] = D.shift[l][2] = 0;
    D.code[l] = IMAGE_NONE;
    for(int q = 0; q < 6; q++) if(brd_mask_of(l) & (1 << q)) {
      off[q / 2] += (q & 1) ? 1 : -1;
      const Swap& sw = h->swaps[q];
      if(sw.pbc_any) for(int d = 0; d < 3; d++) D.shift[l][d] += sw.pbc[d] * h->prd[d];
      D.code[l] = image_add_host(D.code[l], sw.pbc[0], sw.pbc[1], sw.pbc[2]);
    }
    D.target[l] = cart_rank(h->procgrid, h->myloc[0] + off[0], h->myloc[1] + off[1], h->myloc[2] + off[2]);
    D.source[l] = cart_rank(h->procgrid, h->myloc[0] - off[0], h->myloc[1] - off[1], h->myloc[2] - off[2]);
  }
  D.npeer_s = D.npeer_r = 0;
  for(int l = 0; l < 26; l++) { D.lps[l] = -1; D.lpr[l] = -1; }
  for(int l = 0; l < 26; l++) {
    if(D.target[l] == h->me && !forced) continue;
    if(D.lps[l] < 0) {
      D.peer_s[D.npeer_s] = D.target[l];
      for(int m = l; m < 26; m++) if(D.target[m] == D.target[l] && !(D.target[m] == h->me && !forced)) D.lps[m] = D.npeer_s;
      D.npeer_s++;
    }
    if(D.lpr[l] < 0) {
      D.peer_r[D.npeer_r] = D.source[l];
      for(int m = l; m < 26; m++) if(D.source[m] == D.source[l] && !(D.target[m] == h->me && !forced)) D.lpr[m] = D.npeer_r;
      D.npeer_r++;
    }
  }
}

// lists + lengths of this rank on the stream, the lengths of the lists it will receive exchanged with the neighbours, both copied to pinned memory
// (no host synchronisation here with RCCL: the neighbor build's own read-back comes later on the same stream)
static int dh_enqueue(mmd_handle* h)
{
  DirectHalo& D = h->dh;
  D.ready = false; D.pending = false;
  if(!dh_applies(h)) return 0;
  const int nlocal = h->nlocal;
  SelfSwaps W;
  dh_self_swaps(h, W);
  dh_topology(h);
  const int nblk = std::max(1, div_up(nlocal, CP_TILE));
  MMD_TRY(h->flag_tmp.ensure((size_t)BRD_ROWS * nblk + BRD_ROWS + 8, false, h->stream));
  MMD_TRY(h->brd_bits.ensure((size_t)nlocal + 64, false, h->stream));
  MMD_TRY(D.scratch.ensure(64, false, h->stream));
  MMD_TRY(D.counts.ensure(32 * 30, false, h->stream));
  MMD_TRY(D.idx.ensure((size_t)h->nlocal + (size_t)h->nghost + h->nghost / 2 + 4096, false, h->stream));       // (a rank sends about as many images as it receives)
  int* tot = h->flag_tmp.p + (size_t)BRD_ROWS * nblk;
  hipLaunchKernelGGL(k_brd_count, dim3(nblk), dim3(256), 0, h->stream, h->x.p, nlocal, W, h->brd_bits.p, h->flag_tmp.p, nblk, D.scratch.p);
  hipLaunchKernelGGL(k_brd_scan, dim3(BRD_ROWS), dim3(256), 0, h->stream, h->flag_tmp.p, nblk, tot);
  HIP_TRY(hipGetLastError());
  const bool forced = h->opt_force_transport != 0;
  if(h->rccl) {
    hipLaunchKernelGGL(k_dh_lists, dim3(nblk), dim3(256), 0, h->stream, nlocal, h->brd_bits.p, h->flag_tmp.p, nblk, tot, D.idx.p, D.counts.p, (int)std::min<size_t>(D.idx.cap, 0x7fffffff));
    HIP_TRY(hipGetLastError());
    // the lengths travel as ONE message per distinct partner: all 26 words to each (a p2p operation costs ~3 us whatever its size)
    ncclComm_t c = (ncclComm_t)h->rccl;
    NCCL_TRY(ncclGroupStart());
    int seen_t[26], seen_s[26], nt_ = 0, ns_ = 0;
    for(int l = 0; l < 26; l++) {
      if(D.target[l] == h->me && !forced) continue;
      bool dup = false;
      for(int k = 0; k < nt_; k++) dup = dup || seen_t[k] == D.target[l];
      if(!dup) { seen_t[nt_++] = D.target[l]; NCCL_TRY(ncclSend(D.counts.p, 26, ncclInt, D.target[l], c, h->stream)); }
      dup = false;
      for(int k = 0; k < ns_; k++) dup = dup || seen_s[k] == D.source[l];
      if(!dup) { NCCL_TRY(ncclRecv(D.counts.p + 32 * (2 + ns_), 26, ncclInt, D.source[l], c, h->stream)); seen_s[ns_++] = D.source[l]; }
    }
    NCCL_TRY(ncclGroupEnd());
    D.nsrc = ns_;
    for(int k = 0; k < ns_; k++) D.src_rank[k] = seen_s[k];
    // (device layout of counts: [0..25] mine, then one block of 32 per distinct source, from word 64 on)
    HIP_TRY(hipMemcpyAsync(D.h_counts, D.counts.p, (size_t)32 * (2 + ns_) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    D.pending = true;
    return 0;
  }
  D.nsrc = 0;
  // host-staged test transport: the lengths go through the host anyway
  hipLaunchKernelGGL(k_dh_lists, dim3(nblk), dim3(256), 0, h->stream, nlocal, h->brd_bits.p, h->flag_tmp.p, nblk, tot, D.idx.p, D.counts.p, (int)std::min<size_t>(D.idx.cap, 0x7fffffff));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(D.h_counts, D.counts.p, 64 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync_transport(h));
  for(int l = 0; l < 26; l++) {
    if(D.target[l] == h->me && !forced) continue;
    int out = 0;
    const long long got = h->host_sr(h->host_ctx, &D.h_counts[l], sizeof(int), D.target[l], &out, sizeof(int), D.source[l]);
    if(got != (long long)sizeof(int)) { mmd_set_error("direct halo: count exchange failed"); return -1; }
    D.h_counts[32 + l] = out;
  }
  D.pending = true;
  return 0;
}

// the lengths are in pinned memory (the caller synchronised, or read the neighbor build's result words, which were published behind them)
static int dh_finish(mmd_handle* h)
{
  DirectHalo& D = h->dh;
  if(!D.pending) return 0;
  D.pending = false;
  const bool forced = h->opt_force_transport != 0;
  int so = 0, rb = 0;
  for(int l = 0; l < 26; l++) {
    D.ns[l] = D.h_counts[l];
    if(D.target[l] == h->me && !forced) D.nr[l] = D.ns[l];
    else if(D.nsrc > 0) {                 // RCCL: every source sent all its 26 lengths; mine is its list l
      int k = 0;
      while(k < D.nsrc && D.src_rank[k] != D.source[l]) k++;
      D.nr[l] = D.h_counts[32 * (2 + k) + l];
    } else D.nr[l] = D.h_counts[32 + l];
    D.soff[l] = so; so += D.ns[l];
    D.rbase[l] = rb; rb += D.nr[l];
  }
  D.soff[26] = so; D.rbase[26] = rb;
  D.total_send = so;
  // one message per distinct partner, its lists end to end in list order (dh_topology: sender and receiver enumerate partners and lists alike)
  int ps = 0, pr = 0;
  for(int l = 0; l < 26; l++) { D.sdst[l] = -1; D.rsrc[l] = -1; }
  for(int k = 0; k < D.npeer_s; k++) {
    D.peer_soff[k] = ps;
    for(int m = 0; m < 26; m++) if(D.lps[m] == k) { D.sdst[m] = ps; ps += D.ns[m]; }
  }
  for(int k = 0; k < D.npeer_r; k++) {
    D.peer_roff[k] = pr;
    for(int m = 0; m < 26; m++) if(D.lpr[m] == k) { D.rsrc[m] = pr; pr += D.nr[m]; }
  }
  D.peer_soff[D.npeer_s] = ps; D.peer_roff[D.npeer_r] = pr;
  D.total_recv = pr;
  if(rb != h->nghost) { mmd_set_error("direct halo: the 26 lists hold %d ghosts, Comm::borders made %d", rb, h->nghost); return -1; }
  if((size_t)so > D.idx.cap) { mmd_set_error("direct halo: send lists longer than provided for (%d)", so); return -1; }
  D.ready = true;
  D.prev_valid = true;                     // (sizes the fixed messages of the next direct borders)
  for(int l = 0; l < 26; l++) { D.ns_prev[l] = D.ns[l]; D.nr_prev[l] = D.nr[l]; }
  return 0;
}

int mmd_dh_exchange(mmd_handle* h, int what)
{
  DirectHalo& D = h->dh;
  if(!D.ready) { mmd_set_error("direct halo: no plan"); return -1; }
  const bool forced = h->opt_force_transport != 0;
  DhMeta M;
  DhUnpack U;
  for(int l = 0; l < 26; l++) {
    M.soff[l] = D.soff[l];
    const bool self = D.target[l] == h->me && !forced;
    M.self_base[l] = self ? D.rbase[l] : -1;
    M.sdst[l] = D.sdst[l];
    M.sx[l] = D.shift[l][0]; M.sy[l] = D.shift[l][1]; M.sz[l] = D.shift[l][2];
    U.rbase[l] = D.rbase[l]; U.rsrc[l] = D.rsrc[l];
  }
  M.soff[26] = D.soff[26]; U.rbase[26] = D.rbase[26];
  const size_t esz = what == 0 ? sizeof(real4) : sizeof(real);
  const size_t per = what == 0 ? 4 : 1;
  // halo_recv 3: the step loop allows it, the build named the boundary tiles' ghosts by gmap: positions are received behind the ghost slots and stay there
  const bool in_x = what == 0 && h->halo_in_x_allow && D.opt_recv == 3 && D.gmap_live && h->cand_src_ready && h->cand_src_halo && h->rccl != nullptr &&
                    (size_t)D.R + (size_t)D.total_recv <= (size_t)h->nmax;
  if(what == 0) D.x_unpack_pending = in_x;
  if(in_x) h->halo_in_x_steps++;
  const int nsend_remote = D.peer_soff[D.npeer_s];
  MMD_TRY(h->buf_send.ensure(per * (size_t)nsend_remote + 16, false, h->stream));
  if(!in_x) MMD_TRY(h->buf_recv.ensure(per * (size_t)D.total_recv + 16, false, h->stream));
  if(D.total_send) {
    if(what == 0) hipLaunchKernelGGL(k_dh_pack_x, dim3(div_up(D.total_send, 256)), dim3(256), 0, h->stream, h->x.p, D.idx.p, D.total_send, M, (real4*)h->buf_send.p, h->nlocal);
    else hipLaunchKernelGGL(k_dh_pack_f, dim3(div_up(D.total_send, 256)), dim3(256), 0, h->stream, h->fp.p, D.idx.p, D.total_send, M, h->buf_send.p, h->nlocal);
    HIP_TRY(hipGetLastError());
  }
  unsigned char* sbuf = (unsigned char*)h->buf_send.p;
  unsigned char* rbuf = in_x ? (unsigned char*)(h->x.p + D.R) : (unsigned char*)h->buf_recv.p;
  if(h->rccl) {
    h->halo_bytes += (long long)((size_t)nsend_remote * esz);
    ncclComm_t c = (ncclComm_t)h->rccl;
    NCCL_TRY(ncclGroupStart());
    for(int k = 0; k < D.npeer_s; k++) {
      const size_t n = (size_t)(D.peer_soff[k + 1] - D.peer_soff[k]);
      if(n) NCCL_TRY(ncclSend(sbuf + (size_t)D.peer_soff[k] * esz, n * esz, ncclChar, D.peer_s[k], c, h->stream));
    }
    for(int k = 0; k < D.npeer_r; k++) {
      const size_t n = (size_t)(D.peer_roff[k + 1] - D.peer_roff[k]);
      if(n) NCCL_TRY(ncclRecv(rbuf + (size_t)D.peer_roff[k] * esz, n * esz, ncclChar, D.peer_r[k], c, h->stream));
    }
    NCCL_TRY(ncclGroupEnd());
  } else {
    // host-staged test transport: a send to peer k is matched by the receive from the rank that has this rank as ITS k-th target; pairing send k with
    // receive k is a shift pattern every rank follows alike only when both enumerate their partners in the same list order — which they do
    const int np = std::max(D.npeer_s, D.npeer_r);
    for(int k = 0; k < np; k++) {
      const bool hs = k < D.npeer_s, hr = k < D.npeer_r;
      const size_t nsb = hs ? (size_t)(D.peer_soff[k + 1] - D.peer_soff[k]) * esz : 0, nrb = hr ? (size_t)(D.peer_roff[k + 1] - D.peer_roff[k]) * esz : 0;
      MMD_TRY(mmd_transport_sendrecv(h, sbuf + (hs ? (size_t)D.peer_soff[k] * esz : 0), nsb, hs ? D.peer_s[k] : h->me, rbuf + (hr ? (size_t)D.peer_roff[k] * esz : 0), nrb,
                                     hr ? D.peer_r[k] : h->me));
    }
  }
  if(D.total_recv && h->nghost && !in_x) {
    if(what == 0) hipLaunchKernelGGL((k_dh_unpack<real4>), dim3(div_up(h->nghost, 256)), dim3(256), 0, h->stream, h->x.p + h->nlocal, (const real4*)h->buf_recv.p, h->nghost, U);
    else hipLaunchKernelGGL((k_dh_unpack<real>), dim3(div_up(h->nghost, 256)), dim3(256), 0, h->stream, h->fp.p + h->nlocal, (const real*)h->buf_recv.p, h->nghost, U);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Direct borders (round 5): Comm::borders (ref/comm.cpp:700-883) on several ranks as ONE exchange.
// The three dependent rounds of the reference exist because a later dimension forwards the ghosts an earlier one received. But whether a ghost is
// forwarded by swap q is decided by the coordinate of its ROOT in q's dimension against slab bounds the forwarding rank shares with the root's owner
// (they differ in earlier dimensions only) — the owner can decide it itself. So every rank compacts its 26 lists (k_brd_count / k_brd_scan /
// k_dh_lists, exactly the lists of the per-step direct halo), packs {x + shift, type} + the atom's slab bits of every list entry into ONE fixed-size
// message per distinct partner (header = the 26 list lengths; capacity from the previous plan's lengths, derived alike on both sides), all messages
// travel in one ncclGroup, and k_db_unpack lays the received lists end to end in list order — which IS the swap-by-swap ghost order (BrdList) — with
// positions, types, image codes; it also leaves every ghost's slab bits (the six send lists of the swaps are derived from them on demand,
// mmd_comm_sendlists_ensure: nothing on the step path reads them) and the swaps' counts in bst. A count beyond its capacity raises the overflow flag
// (max-reduced over the ranks): every rank then redoes the borders swap by swap. Six launches + one RCCL group instead of 14 launches + 3 groups, one
// dependent transfer instead of three, and the direct-halo plan of the steps comes out of the same lists (no second pass, no separate length exchange).
// ---------------------------------------------------------------------------------------------------
static int borders_fast_finish(mmd_handle* h);
#define DBMSG_HEADER 128
static inline size_t db_msg_bytes(int cap) { return ((size_t)DBMSG_HEADER + (size_t)cap * (sizeof(real4) + sizeof(int)) + 63) & ~(size_t)63; }
struct DbPack {
  int lps[26];                  // outgoing message of list l (self lists: the extra message np_s)
  int cap[27];                  // record capacity of message m
  unsigned char* msg[27];
  int nmsg;
  real sx[26], sy[26], sz[26];
  int code[26];
  int idx_cap;
};
__global__ __launch_bounds__(256) void k_db_pack(const real4* __restrict__ x, const int* __restrict__ idx, const int* __restrict__ counts,
                                                 const unsigned char* __restrict__ bits, DbPack P, int* __restrict__ bst)
{
  __shared__ int s_soff[27], s_poff[26], s_tot[27];
  if(threadIdx.x == 0) {
    int so = 0;
    for(int m = 0; m < 27; m++) s_tot[m] = 0;
    for(int l = 0; l < 26; l++) {
      const int n = counts[l], m = P.lps[l];
      s_soff[l] = so; so += n;
      s_poff[l] = s_tot[m]; s_tot[m] += n;
    }
    s_soff[26] = so;
  }
  __syncthreads();
  if(blockIdx.x == 0 && threadIdx.x < P.nmsg) {          // headers: my 26 list lengths (the receiver reads those of the lists it gets from me), records in this message
    int* hd = (int*)P.msg[threadIdx.x];
    for(int l = 0; l < 26; l++) hd[l] = counts[l];
    hd[26] = s_tot[threadIdx.x];
    hd[27] = 0x6d6d6462;
    if(s_tot[threadIdx.x] > P.cap[threadIdx.x]) bst[BST_OVF] = 1;
  }
  if(blockIdx.x == 0 && threadIdx.x == 0 && s_soff[26] > P.idx_cap) bst[BST_OVF] = 1;       // (k_dh_lists dropped entries)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= s_soff[26] || e >= P.idx_cap) return;
  int l = 0;
#pragma unroll
  for(int q = 1; q < 26; q++) l += e >= s_soff[q] ? 1 : 0;
  const int m = P.lps[l], o = s_poff[l] + (e - s_soff[l]);
  if(o >= P.cap[m]) return;                               // (overflow: flagged above)
  const int i = idx[e];
  real4 p = x[i];
  p.x += P.sx[l]; p.y += P.sy[l]; p.z += P.sz[l];       // (a list without periodic swaps carries zeros: x + 0 is x)
  real4* rec = (real4*)(P.msg[m] + DBMSG_HEADER);
  int* aux = (int*)(P.msg[m] + DBMSG_HEADER + (size_t)P.cap[m] * sizeof(real4));
  rec[o] = p;
  aux[o] = (int)bits[i] | (P.code[l] << 8);
}
struct DbUnpack {
  int lpr[26];                  // incoming message of list l
  int cap[27];
  const unsigned char* msg[27];
  int nmsg;
  int lpr_step[26];             // per-step halo: partner message list l arrives in (-1: the list stays on this rank and is written into its ghost slots)
  int R;                        // first entry of the position buffer behind the ghost slots (0: no gmap wanted)
};
__global__ __launch_bounds__(256) void k_db_unpack(real4* __restrict__ x, int nlocal, DbUnpack U, const int* __restrict__ counts, const int* __restrict__ tot_any,
                                                   int cap_atoms, int cap_ghost, int* __restrict__ ghost_image, unsigned char* __restrict__ ghost_bits,
                                                   int* __restrict__ type, int* __restrict__ bst, int* __restrict__ host_counts, int* __restrict__ gmap)
{
  __shared__ int s_nr[26], s_rbase[27], s_roff[26], s_goff[26];
  if(threadIdx.x == 0) {
    int ovf = 0, rb = 0, mt[27];
    for(int m = 0; m < 27; m++) mt[m] = 0;
    for(int m = 0; m < U.nmsg; m++) { const int* hd = (const int*)U.msg[m]; if(hd[27] != 0x6d6d6462 || hd[26] > U.cap[m] || hd[26] < 0) ovf = 1; }
    for(int l = 0; l < 26; l++) {
      const int m = U.lpr[l];
      int n = ((const int*)U.msg[m])[l];
      n = min(max(n, 0), max(U.cap[m] - mt[m], 0));       // (never read beyond a message, whatever its header says)
      s_nr[l] = n; s_rbase[l] = rb; s_roff[l] = mt[m];
      rb += n; mt[m] += n;
    }
    s_rbase[26] = rb;
    if(rb > cap_ghost || nlocal + rb + 1 > cap_atoms || (U.R > 0 && (nlocal + rb + 1 > U.R || U.R + rb > cap_atoms))) ovf = 1;
    if(blockIdx.x == 0) {
      if(ovf) bst[BST_OVF] = 1;
      int first = 0;
      int rq[6] = {0, 0, 0, 0, 0, 0};
      for(int l = 0; l < 26; l++) { const int q = l < 1 ? 0 : l < 2 ? 1 : l < 5 ? 2 : l < 8 ? 3 : l < 17 ? 4 : 5; rq[q] += s_nr[l]; }
      for(int q = 0; q < 6; q++) { bst[BST_RECV + q] = rq[q]; bst[BST_GHOSTS + q] = first; first += rq[q]; }
      bst[BST_GHOSTS + 6] = rb;
      bst[BST_NB] = tot_any[0];
      // what this rank's swaps send: the owned atoms inside the swap's slab (its one-swap list) + the forwarded ghosts counted below
      atomicAdd(&bst[BST_SEND + 0], counts[0]); atomicAdd(&bst[BST_SEND + 1], counts[1]); atomicAdd(&bst[BST_SEND + 2], counts[2]);
      atomicAdd(&bst[BST_SEND + 3], counts[5]); atomicAdd(&bst[BST_SEND + 4], counts[8]); atomicAdd(&bst[BST_SEND + 5], counts[17]);
      for(int l = 0; l < 26; l++) { host_counts[l] = counts[l]; host_counts[32 + l] = s_nr[l]; }
    }
  }
  __syncthreads();
  if(threadIdx.x < 26) {         // where list l starts in the per-step receive layout: the partners' messages in partner order, a partner's lists end to end
    const int l = threadIdx.x, k = U.lpr_step[l];
    int o = 0;
    for(int m = 0; m < 26; m++) { const int km = U.lpr_step[m]; if(km >= 0 && (km < k || (km == k && m < l))) o += s_nr[m]; }
    s_goff[l] = k < 0 ? -1 : o;
  }
  __syncthreads();
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int nghost = min(s_rbase[26], min(cap_ghost, cap_atoms - nlocal - 1));
  int F = 0, l = 0;
  if(g < nghost) {
#pragma unroll
    for(int q = 1; q < 26; q++) l += g >= s_rbase[q] ? 1 : 0;
    const int m = U.lpr[l], o = s_roff[l] + (g - s_rbase[l]);
    const real4* rec = (const real4*)(U.msg[m] + DBMSG_HEADER);
    const int* aux = (const int*)(U.msg[m] + DBMSG_HEADER + (size_t)U.cap[m] * sizeof(real4));
    const real4 p = rec[o];
    const int a = aux[o];
    x[nlocal + g] = p;
    type[nlocal + g] = (int)p.w;
    ghost_image[g] = a >> 8;
    ghost_bits[g] = (unsigned char)(a & 0xff);
    if(U.R > 0) gmap[g] = s_goff[l] < 0 ? nlocal + g : U.R + s_goff[l] + (g - s_rbase[l]);
    // a swap of dimension d forwards the ghosts of the EARLIER dimensions that lie in its slab (lists 0-1: x, 2-7: up to y)
    F = a & 0xff;
    if(l >= 8) F = 0; else if(l >= 2) F &= 0x30; else F &= 0x3c;
  }
#pragma unroll
  for(int q = 2; q < 6; q++) {
    const int c = __popcll(__builtin_amdgcn_ballot_w64((F >> q) & 1));
    if(c && (threadIdx.x & 63) == 0) atomicAdd(&bst[BST_SEND + q], c);
  }
}

struct BitsPred {      // send list of swap q from the slab bits: owned atoms and ghosts of the earlier dimensions inside its slab, ascending (ref/comm.cpp:766-779)
  const unsigned char* own; const unsigned char* ghost; int nlocal, q;
  __device__ bool operator()(int i) const { return (((int)(i < nlocal ? own[i] : ghost[i - nlocal])) >> q) & 1; }
  __device__ int index(int i) const { return i; }
};
int mmd_comm_sendlists_ensure(mmd_handle* h)
{
  if(!h->sendlists_stale) return 0;
  for(int q = 0; q < 6; q++) {
    Swap& s = h->swaps[q];
    const int nlast = q < 2 ? h->nlocal : h->swaps[q & ~1].firstrecv;       // (both swaps of a dimension scan the atoms in front of the dimension's first ghost)
    // (the length is known — k_db_unpack counted it into bst — so the compaction needs no read-back: count, scan, scatter on the stream)
    if(nlast <= 0 || s.sendnum <= 0) continue;
    const BitsPred pred{h->brd_bits.p, h->ghost_bits.p, h->nlocal, q};
    const int ntiles = div_up(nlast, CP_TILE);
    MMD_TRY(h->flag_tmp.ensure((size_t)ntiles + 8, false, h->stream));
    MMD_TRY(s.sendlist.ensure((size_t)s.sendnum + 8, false, h->stream));
    hipLaunchKernelGGL((k_compact_count<BitsPred>), dim3(ntiles), dim3(256), 0, h->stream, pred, 0, nlast, h->flag_tmp.p);
    MMD_TRY(mmd_exclusive_scan(h, h->flag_tmp.p, ntiles, nullptr));
    hipLaunchKernelGGL((k_compact_scatter<BitsPred>), dim3(ntiles), dim3(256), 0, h->stream, pred, 0, nlast, h->flag_tmp.p, s.sendlist.p);
    HIP_TRY(hipGetLastError());
  }
  h->sendlists_stale = false;
  return 0;
}

// 2 = enqueued (deferred: counts still on the device), 1 = done, 0 = not applicable / overflow (the caller runs the swap-by-swap path), < 0 error
static int borders_direct(mmd_handle* h, bool defer)
{
  DirectHalo& D = h->dh;
  if(!D.opt_borders || !D.prev_valid || !dh_applies(h) || !h->borders_general_done) return 0;
  const bool forced = h->opt_force_transport != 0;
  const int nlocal = h->nlocal;
  dh_topology(h);
  SelfSwaps W;
  dh_self_swaps(h, W);
  // fixed-size messages: message k of a side holds the lists of partner k; both sides size it from the same previous lengths
  DbPack P;
  DbUnpack U;
  int caps_s[27], caps_r[27];
  size_t off_s[28], off_r[28];
  const int np_s = D.npeer_s, np_r = D.npeer_r;
  bool any_self = false;
  for(int l = 0; l < 26; l++) any_self = any_self || D.lps[l] < 0;
  for(int k = 0; k <= 26; k++) { caps_s[k] = caps_r[k] = 0; }
  for(int l = 0; l < 26; l++) {
    caps_s[D.lps[l] < 0 ? np_s : D.lps[l]] += D.ns_prev[l];
    caps_r[D.lpr[l] < 0 ? np_r : D.lpr[l]] += D.nr_prev[l];
  }
  const int nmsg_s = np_s + (any_self ? 1 : 0), nmsg_r = np_r + (any_self ? 1 : 0);
  size_t bs = 0, br = 0;
  long long cap_send_total = 0, cap_recv_total = 0;
  // (opt_borders_est < 100: tests shrink the capacities to force the overflow protocol)
  auto border_msg_cap = [h](int prev) { return h->opt_borders_est >= 100 ? ::border_msg_cap(prev) : (int)((long long)prev * h->opt_borders_est / 100); };
  for(int k = 0; k < nmsg_s; k++) { caps_s[k] = border_msg_cap(caps_s[k]); off_s[k] = bs; bs += db_msg_bytes(caps_s[k]); cap_send_total += caps_s[k]; }
  for(int k = 0; k < np_r; k++) { caps_r[k] = border_msg_cap(caps_r[k]); off_r[k] = br; br += db_msg_bytes(caps_r[k]); cap_recv_total += caps_r[k]; }
  if(any_self) { caps_r[np_r] = caps_s[np_s]; cap_recv_total += caps_r[np_r]; }      // (the lists that stay here are read where k_db_pack wrote them)
  if(cap_send_total > 0x3fffffff || cap_recv_total > 0x3fffffff) return 0;
  const int est_ghost = (int)cap_recv_total;
  // halo_recv 3: the per-step halo of the partners lands behind the ghost slots of the position buffer (DirectHalo::gmap)
  const bool want_gmap = D.opt_recv == 3 && h->style == 0 && !h->halfneigh && h->opt_tiles && h->lj_uniform && h->opt_ghost_resolve &&
                         (long long)nlocal + 2 * (long long)est_ghost + 16 < (1 << MMD_SRC_BITS);
  const int R = want_gmap ? ((nlocal + est_ghost + 1 + 3) & ~3) : 0;
  MMD_TRY(mmd_ensure_atoms(h, want_gmap ? R + est_ghost + 1 : nlocal + est_ghost + 1, true));
  if(want_gmap) MMD_TRY(D.gmap.ensure((size_t)est_ghost + 8, false, h->stream));
  MMD_TRY(h->ghost_image.ensure((size_t)est_ghost + 8, false, h->stream));
  MMD_TRY(h->ghost_bits.ensure((size_t)est_ghost + 8, false, h->stream));
  MMD_TRY(h->buf_send.ensure(bs / sizeof(real) + 16, false, h->stream));
  MMD_TRY(h->buf_recv.ensure(br / sizeof(real) + 16, false, h->stream));
  MMD_TRY(h->bstate.ensure(64, false, h->stream));
  const int nblk = std::max(1, div_up(nlocal, CP_TILE));
  MMD_TRY(h->flag_tmp.ensure((size_t)BRD_ROWS * nblk + BRD_ROWS + 8, false, h->stream));
  MMD_TRY(h->brd_bits.ensure((size_t)nlocal + 64, false, h->stream));
  MMD_TRY(D.counts.ensure(32 * 30, false, h->stream));
  MMD_TRY(D.idx.ensure((size_t)cap_send_total + 64, false, h->stream));
  unsigned char* sbuf = (unsigned char*)h->buf_send.p;
  unsigned char* rbuf = (unsigned char*)h->buf_recv.p;
  for(int l = 0; l < 26; l++) {
    P.lps[l] = D.lps[l] < 0 ? np_s : D.lps[l];
    U.lpr[l] = D.lpr[l] < 0 ? np_r : D.lpr[l];
    U.lpr_step[l] = D.lpr[l];
    P.sx[l] = D.shift[l][0]; P.sy[l] = D.shift[l][1]; P.sz[l] = D.shift[l][2];
    P.code[l] = D.code[l];
  }
  for(int k = 0; k < 27; k++) { P.cap[k] = caps_s[k]; U.cap[k] = caps_r[k]; P.msg[k] = nullptr; U.msg[k] = nullptr; }
  for(int k = 0; k < nmsg_s; k++) P.msg[k] = sbuf + off_s[k];
  for(int k = 0; k < np_r; k++) U.msg[k] = rbuf + off_r[k];
  if(any_self) U.msg[np_r] = sbuf + off_s[np_s];
  P.nmsg = nmsg_s; U.nmsg = nmsg_r;
  U.R = R;
  D.R = R; D.gmap_live = want_gmap; D.x_unpack_pending = false;
  P.idx_cap = (int)std::min<size_t>(D.idx.cap, 0x7fffffff);
  int* tot = h->flag_tmp.p + (size_t)BRD_ROWS * nblk;
  hipLaunchKernelGGL(k_brd_count, dim3(nblk), dim3(256), 0, h->stream, h->x.p, nlocal, W, h->brd_bits.p, h->flag_tmp.p, nblk, h->bstate.p);
  hipLaunchKernelGGL(k_brd_scan, dim3(BRD_ROWS), dim3(256), 0, h->stream, h->flag_tmp.p, nblk, tot);
  hipLaunchKernelGGL(k_dh_lists, dim3(nblk), dim3(256), 0, h->stream, nlocal, h->brd_bits.p, h->flag_tmp.p, nblk, tot, D.idx.p, D.counts.p, P.idx_cap);
  hipLaunchKernelGGL(k_db_pack, dim3(std::max(1, div_up(cap_send_total, 256))), dim3(256), 0, h->stream, h->x.p, D.idx.p, D.counts.p, h->brd_bits.p, P, h->bstate.p);
  HIP_TRY(hipGetLastError());
  if(h->rccl) {
    ncclComm_t c = (ncclComm_t)h->rccl;
    NCCL_TRY(ncclGroupStart());
    for(int k = 0; k < np_s; k++) { const size_t n = db_msg_bytes(caps_s[k]); h->halo_bytes += (long long)n; NCCL_TRY(ncclSend(sbuf + off_s[k], n, ncclChar, D.peer_s[k], c, h->stream)); }
    for(int k = 0; k < np_r; k++) NCCL_TRY(ncclRecv(rbuf + off_r[k], db_msg_bytes(caps_r[k]), ncclChar, D.peer_r[k], c, h->stream));
    NCCL_TRY(ncclGroupEnd());
  } else {
    const int np = std::max(np_s, np_r);        // (send k paired with receive k: the shift pattern of mmd_dh_exchange's host path)
    for(int k = 0; k < np; k++) {
      const bool hs = k < np_s, hr = k < np_r;
      MMD_TRY(mmd_transport_sendrecv(h, sbuf + (hs ? off_s[k] : 0), hs ? db_msg_bytes(caps_s[k]) : 0, hs ? D.peer_s[k] : h->me, rbuf + (hr ? off_r[k] : 0),
                                     hr ? db_msg_bytes(caps_r[k]) : 0, hr ? D.peer_r[k] : h->me));
    }
  }
  const int cap_atoms = h->nmax, cap_ghost = (int)std::min<size_t>(std::min(h->ghost_image.cap, h->ghost_bits.cap), 0x7fffffff);
  hipLaunchKernelGGL(k_db_unpack, dim3(std::max(1, div_up(cap_recv_total, 256))), dim3(256), 0, h->stream, h->x.p, nlocal, U, (const int*)D.counts.p, (const int*)(tot + BRD_NL),
                     cap_atoms, cap_ghost, h->ghost_image.p, h->ghost_bits.p, h->type.p, h->bstate.p, D.h_counts_dev, D.gmap.p);
  HIP_TRY(hipGetLastError());
  MMD_TRY(reduce_flag_max(h, h->bstate.p + BST_OVF));
  h->bf_est_nb = 0x7fffffff;
  D.nsrc = 0; D.pending = true; D.ready = false;
  (void)forced;
  h->borders_direct_pending = true;
  if(defer) {
    h->nghost = std::min(cap_ghost, cap_atoms - nlocal - 1);
    h->nghost = std::min(h->nghost, est_ghost);
    h->nghost_dev = h->bstate.p + BST_GHOSTS + 6;
    for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x.p) h->xalt_dummy_slot[k] = -1;
    return 2;
  }
  HIP_TRY(hipMemcpyAsync(h->h_flags_big, h->bstate.p, 40 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return borders_fast_finish(h);
}

// the ghost slots of x in step with their owners again: after steps whose force kernels staged the ghosts themselves (one rank: from the owners;
// several ranks with halo_recv 3: from the received messages behind the ghost slots — those are copied into the slots, nothing travels again)
int mmd_ghosts_refresh(mmd_handle* h)
{
  DirectHalo& D = h->dh;
  if(D.x_unpack_pending && D.ready) {
    D.x_unpack_pending = false;
    if(D.total_recv && h->nghost) {
      DhUnpack U;
      for(int l = 0; l < 26; l++) { U.rbase[l] = D.rbase[l]; U.rsrc[l] = D.rsrc[l]; }
      U.rbase[26] = D.rbase[26];
      hipLaunchKernelGGL((k_dh_unpack<real4>), dim3(div_up(h->nghost, 256)), dim3(256), 0, h->stream, h->x.p + h->nlocal, (const real4*)(h->x.p + D.R), h->nghost, U);
      HIP_TRY(hipGetLastError());
    }
    return 0;
  }
  return mmd_comm_communicate(h);
}

// returns 1 when the fast path produced the ghosts, 0 when the caller must run the general path
static int borders_fast_finish(mmd_handle* h);

static int borders_device_resident(mmd_handle* h, bool defer)
{
  if(h->swaps.size() != 6) return 0;
  // several ranks: every condition below is the same on all of them (options, topology, "a swap-by-swap borders has run before") — a
  // rank that owned no boundary atom or got no ghost last time (prev_nb, prev_nghost = 0: empty or sparse sub-domain) must still post the
  // fixed-size messages its partners wait for; its arrays get the +4096 floor of the estimates
  if(h->nprocs > 1 ? !h->borders_general_done : (h->prev_nghost <= 0 || h->prev_nb <= 0)) return 0;
  if(h->nprocs == 1 && !h->opt_force_transport && h->nlocal <= 4096) return 0;
  // (force_transport, a test option: periodic self swaps take the message path too — the RCCL calls of this function run on one GPU)
  const bool forced = h->opt_force_transport != 0;
  bool any_remote = false;
  for(int q = 0; q < 6; q += 2) {
    const bool r0 = forced || h->swaps[q].sendproc != h->me, r1 = forced || h->swaps[q + 1].sendproc != h->me;
    if(r0 != r1) return 0;                      // (a dimension is either all-self or all-remote)
    any_remote = any_remote || r0;
  }
  if(any_remote && !(h->rccl || h->host_sr)) return 0;
  const int nlocal = h->nlocal;
  // (opt_borders_est: per cent of the previous counts the arrays are sized for; tests shrink it to force the overflow fallback)
  const int est_ghost = h->opt_borders_est >= 100 ? (int)((long long)h->prev_nghost * h->opt_borders_est / 100) + 4096 : (int)((long long)h->prev_nghost * h->opt_borders_est / 100);
  const int est_nb = h->prev_nb + h->prev_nb / 2 + 4096;
  MMD_TRY(mmd_ensure_atoms(h, nlocal + est_ghost + 1, true));
  MMD_TRY(h->ghost_image.ensure((size_t)est_ghost + 8, false, h->stream));
  MMD_TRY(h->ghost_root.ensure((size_t)est_ghost + 8, false, h->stream));
  MMD_TRY(h->bnd_list.ensure((size_t)nlocal + 8, false, h->stream));
  int cap_list[6];
  for(int q = 0; q < 6; q++) {
    const int est = h->swaps[q].sendnum + h->swaps[q].sendnum / 2 + 4096;
    MMD_TRY(h->swaps[q].sendlist.ensure((size_t)est, false, h->stream));
    cap_list[q] = (int)std::min<size_t>(h->swaps[q].sendlist.cap, 0x7fffffff);
  }
  int cap_atoms = h->nmax, cap_ghost = (int)std::min<size_t>(std::min(h->ghost_image.cap, h->ghost_root.cap), 0x7fffffff);
  if(h->opt_borders_est < 100) { cap_ghost = std::min(cap_ghost, est_ghost); cap_atoms = std::min(cap_atoms, nlocal + est_ghost); }
  const int cap_ghost_eff = std::min(cap_ghost, cap_atoms - nlocal - 1);     // ghosts the scatter kernels can place at most
  const int nt_own = div_up(nlocal, CP_TILE), nt_sw = div_up(est_nb + est_ghost, CP_TILE);
  MMD_TRY(h->flag_tmp.ensure((size_t)std::max(nt_own, 2 * nt_sw) + 8, false, h->stream));
  MMD_TRY(h->bstate.ensure(64, false, h->stream));
  bool fused = !any_remote;
  for(int q = 0; q < 6 && fused; q++) fused = h->swaps[q].dim == q / 2;
  if(fused) {
    // every swap a periodic self swap: count / scan / scatter over the 26 image lists (above)
    SelfSwaps W;
    for(int q = 0; q < 6; q++) {
      const Swap& sw = h->swaps[q];
      W.lo[q] = sw.slablo; W.hi[q] = sw.slabhi;
      W.sx[q] = sw.pbc[0] * h->prd[0]; W.sy[q] = sw.pbc[1] * h->prd[1]; W.sz[q] = sw.pbc[2] * h->prd[2];
      W.pbc_any[q] = sw.pbc_any; W.px[q] = sw.pbc[0]; W.py[q] = sw.pbc[1]; W.pz[q] = sw.pbc[2];
      W.cap_list[q] = cap_list[q]; W.sendlist[q] = h->swaps[q].sendlist.p;
    }
    const int nblk = nt_own > 0 ? nt_own : 1;
    MMD_TRY(h->flag_tmp.ensure((size_t)BRD_ROWS * nblk + BRD_ROWS + 8, false, h->stream));
    MMD_TRY(h->brd_bits.ensure((size_t)nlocal + 64, false, h->stream));
    int* tot = h->flag_tmp.p + (size_t)BRD_ROWS * nblk;
    hipLaunchKernelGGL(k_brd_count, dim3(nblk), dim3(256), 0, h->stream, h->x.p, nlocal, W, h->brd_bits.p, h->flag_tmp.p, nblk, h->bstate.p);
    hipLaunchKernelGGL(k_brd_scan, dim3(BRD_ROWS), dim3(256), 0, h->stream, h->flag_tmp.p, nblk, tot);
    hipLaunchKernelGGL(k_brd_scatter, dim3(nblk), dim3(256), 0, h->stream, h->x.p, nlocal, W, h->brd_bits.p, h->flag_tmp.p, nblk, tot, h->bstate.p,
                       cap_atoms, cap_ghost_eff, h->ghost_image.p, h->ghost_root.p, h->type.p);
    HIP_TRY(hipGetLastError());
    h->bf_est_nb = 0x7fffffff;                   // (no launch of this form is sized by the number of boundary atoms)
    if(defer) {
      h->nghost = cap_ghost_eff;
      h->nghost_dev = h->bstate.p + BST_GHOSTS + 6;
      for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x.p) h->xalt_dummy_slot[k] = -1;
      return 2;
    }
    HIP_TRY(hipMemcpyAsync(h->h_flags_big, h->bstate.p, 40 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    return borders_fast_finish(h);
  }
  SlabSet S;
  S.n = 6;
  for(int q = 0; q < 6; q++) { S.lo[q] = h->swaps[q].slablo; S.hi[q] = h->swaps[q].slabhi; S.dim[q] = h->swaps[q].dim; }
  hipLaunchKernelGGL(k_bnd_count, dim3(nt_own > 0 ? nt_own : 1), dim3(256), 0, h->stream, h->x.p, nlocal, S, h->flag_tmp.p, h->bstate.p);
  hipLaunchKernelGGL(k_bnd_scatter, dim3(nt_own > 0 ? nt_own : 1), dim3(256), 0, h->stream, h->x.p, nlocal, S, h->flag_tmp.p, h->bnd_list.p, h->bstate.p, est_nb);
  for(int q = 0; q < 6; q += 2) {
    const bool remote = forced || h->swaps[q].sendproc != h->me;
    SwapPair P;
    int cap_s[2] = {0, 0}, cap_r[2] = {0, 0};
    size_t off_s[2] = {0, 0}, off_r[2] = {0, 0}, bytes_s[2] = {0, 0}, bytes_r[2] = {0, 0};
    if(remote) {
      // fixed-size messages: both sides derive the capacity of a swap's message from its count at the previous re-neighboring
      for(int y = 0; y < 2; y++) {
        cap_s[y] = border_msg_cap(h->swaps[q + y].sendnum); cap_r[y] = border_msg_cap(h->swaps[q + y].recvnum);
        bytes_s[y] = border_msg_bytes(cap_s[y]); bytes_r[y] = border_msg_bytes(cap_r[y]);
      }
      off_s[1] = bytes_s[0]; off_r[1] = bytes_r[0];
      MMD_TRY(h->buf_send.ensure((bytes_s[0] + bytes_s[1]) / sizeof(real) + 16, false, h->stream));
      MMD_TRY(h->buf_recv.ensure((bytes_r[0] + bytes_r[1]) / sizeof(real) + 16, false, h->stream));
    }
    for(int y = 0; y < 2; y++) {
      Swap& sw = h->swaps[q + y];
      P.lo[y] = sw.slablo; P.hi[y] = sw.slabhi;
      P.sx[y] = sw.pbc[0] * h->prd[0]; P.sy[y] = sw.pbc[1] * h->prd[1]; P.sz[y] = sw.pbc[2] * h->prd[2];
      P.pbc_any[y] = sw.pbc_any; P.px[y] = sw.pbc[0]; P.py[y] = sw.pbc[1]; P.pz[y] = sw.pbc[2];
      P.cap_list[y] = cap_list[q + y]; P.sendlist[y] = sw.sendlist.p;
      P.msg[y] = remote ? (unsigned char*)h->buf_send.p + off_s[y] : nullptr; P.cap_msg[y] = cap_s[y];
    }
    const int dim = h->swaps[q].dim;
    hipLaunchKernelGGL(k_swap_count, dim3(nt_sw, 2), dim3(256), 0, h->stream, h->x.p, h->bnd_list.p, h->bstate.p, nlocal, q, dim, P, h->flag_tmp.p, nt_sw);
    if(!remote) {
      hipLaunchKernelGGL(k_swap_scatter<false>, dim3(nt_sw, 2), dim3(256), 0, h->stream, h->x.p, h->bnd_list.p, h->bstate.p, nlocal, q, dim, P,
                         h->flag_tmp.p, nt_sw, cap_atoms, cap_ghost, h->ghost_image.p, h->ghost_root.p, h->type.p);
      continue;
    }
    hipLaunchKernelGGL(k_swap_scatter<true>, dim3(nt_sw, 2), dim3(256), 0, h->stream, h->x.p, h->bnd_list.p, h->bstate.p, nlocal, q, dim, P,
                       h->flag_tmp.p, nt_sw, cap_atoms, cap_ghost, h->ghost_image.p, h->ghost_root.p, h->type.p);
    HIP_TRY(hipGetLastError());
    const void* dsend[2] = {P.msg[0], P.msg[1]};
    void* drecv[2] = {(unsigned char*)h->buf_recv.p + off_r[0], (unsigned char*)h->buf_recv.p + off_r[1]};
    const int dest[2] = {h->swaps[q].sendproc, h->swaps[q + 1].sendproc}, src[2] = {h->swaps[q].recvproc, h->swaps[q + 1].recvproc};
    MMD_TRY(mmd_transport_sendrecv_pair(h, dsend, bytes_s, dest, drecv, bytes_r, src));
    hipLaunchKernelGGL(k_border_unpack, dim3(div_up(std::max(std::max(cap_r[0], cap_r[1]), 1), 256), 2), dim3(256), 0, h->stream, h->x.p, h->bstate.p, nlocal, q,
                       (const unsigned char*)drecv[0], (const unsigned char*)drecv[1], cap_r[0], cap_r[1], cap_atoms, cap_ghost, h->ghost_image.p, h->type.p);
  }
  HIP_TRY(hipGetLastError());
  if(any_remote) {
    // the overflow flag is made global before anybody reads it: either every rank keeps these ghosts or every rank redoes the
    // borders swap by swap (the general path is a sequence of matched sends and receives)
    MMD_TRY(reduce_flag_max(h, h->bstate.p + BST_OVF));
  }
  static_assert(BST_GHOSTS + 6 < 40, "bst read-back window");
  h->bf_est_nb = est_nb;
  if(defer) {
    // inside a re-neighboring the counts are not needed on the host before the neighbor build has run: the kernels up to there read
    // the ghost count from bst (deferred_count), the build's own read-back brings bst along (mmd_borders_deferred_finish)
    h->nghost = cap_ghost_eff;                                   // (a bound, for array sizes and grids only)
    h->nghost_dev = h->bstate.p + BST_GHOSTS + 6;
    // (the dummy atom behind the last ghost is written by k_pencil_fill of the neighbor build that follows)
    for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x.p) h->xalt_dummy_slot[k] = -1;
    return 2;
  }
  HIP_TRY(hipMemcpyAsync(h->h_flags_big, h->bstate.p, 40 * sizeof(int), hipMemcpyDeviceToHost, h->stream));      // nb, ovf, sendnum[6], ghost prefix
  HIP_TRY(mmd_stream_sync(h));
  return borders_fast_finish(h);
}

// host side of the one-rank fast path, from the bst copy in h_flags_big: 1 = done, 0 = estimates too small (general path must redo it)
static int borders_fast_finish(mmd_handle* h)
{
  const int nlocal = h->nlocal;
  const int* hf = h->h_flags_big;
  const bool direct = h->borders_direct_pending;
  h->borders_direct_pending = false;
  if(hf[BST_OVF] || hf[BST_NB] > h->bf_est_nb) { if(direct) { h->dh.pending = false; h->dh.prev_valid = false; h->dh.gmap_live = false; } return 0; }        // estimates too small: general path (it grows the arrays)
  int nall = nlocal;
  bool all_self = true;
  for(int q = 0; q < 6; q++) {
    Swap& s = h->swaps[q];
    s.sendnum = hf[BST_SEND + q];
    s.recvnum = hf[BST_RECV + q];
    s.firstrecv = nall;
    nall += s.recvnum;
    all_self = all_self && s.sendproc == h->me && !h->opt_force_transport;
  }
  h->nghost = hf[BST_GHOSTS + 6];
  if(nall != nlocal + h->nghost) { mmd_set_error("borders fast path: inconsistent ghost counts"); return -1; }
  h->prev_nb = hf[BST_NB];
  h->prev_nghost = h->nghost;
  h->ghost_chain_ok = all_self;
  h->sendlists_stale = direct;               // (direct borders: the swaps' lists are derived from the slab bits when somebody asks)
  if(direct) h->borders_direct_runs++; else h->borders_fast_runs++;
  return 1;
}

static int borders_general(mmd_handle* h);

// Everything the re-neighborings of a run will ask for, allocated before the run's clock starts (mmd_integrate_run): the device-resident borders sizes
// its arrays from the previous counts + 50 %, Atom::sort and the fused integrator want second copies of the per-atom arrays — left to their first use they
// are (re)allocated inside the first two re-neighborings, and a DevArr that grows waits for the stream and frees (670 + 40 us of idle GPU in the
// first 40 steps of `miniMD -s 80`: 3 % of the 100-step PERF_SUMMARY run). Sizes: what those paths would request, with room for the counts to
// grow by a third (a melting lattice: +15 % boundary atoms between the first two re-neighborings). Growing later still works, it only costs.
int mmd_run_reserve(mmd_handle* h)
{
  if(h->host_only || h->nlocal <= 0) return 0;
  const int nlocal = h->nlocal, nghost = std::max(h->nghost, h->prev_nghost);
  const size_t est_ghost = (size_t)2 * nghost + 8192;
  MMD_TRY(h->ghost_image.ensure(est_ghost + 8, true, h->stream));
  MMD_TRY(h->ghost_root.ensure(est_ghost + 8, true, h->stream));
  MMD_TRY(h->bnd_list.ensure((size_t)nlocal + 8, false, h->stream));
  for(auto& sw : h->swaps) MMD_TRY(sw.sendlist.ensure((size_t)2 * sw.sendnum + 8192, true, h->stream));
  const int nt_own = div_up(nlocal, CP_TILE), nt_sw = div_up((long long)h->prev_nb * 2 + est_ghost + 8192, CP_TILE);
  MMD_TRY(h->flag_tmp.ensure((size_t)std::max({nt_own, 2 * nt_sw, BRD_ROWS * std::max(nt_own, 1) + BRD_ROWS}) + 8, false, h->stream));
  MMD_TRY(h->bstate.ensure(64, false, h->stream));
  MMD_TRY(h->brd_bits.ensure((size_t)nlocal + 64, false, h->stream));
  // second copies: Atom::sort (all four), the fused force + integrate kernels (positions)
  const size_t need = (size_t)h->nmax + 1;
  MMD_TRY(h->x_alt.ensure(need, false, h->stream));
  if(h->sort_every > 0) {
    MMD_TRY(h->v_alt.ensure(3 * need, false, h->stream));
    MMD_TRY(h->type_alt.ensure(need, false, h->stream));
    MMD_TRY(h->tag_alt.ensure(need, false, h->stream));
  }
  // the binning of a build behind a deferred borders covers owned atoms + the ghost CAPACITY (the count is still on the device)
  MMD_TRY(h->atom_bin.ensure(need, false, h->stream));
  MMD_TRY(h->atom_rank.ensure(need, false, h->stream));
  MMD_TRY(h->binned.ensure(need, true, h->stream));        // (live: the tiles of the current lists name their atoms through it)
  if(h->neigh_ready) MMD_TRY(h->bin_start_alt.ensure((size_t)h->bg.mbins + 8, false, h->stream));
  if(h->neigh_ready) MMD_TRY(h->pencil_lohi.ensure((size_t)2 * h->bg.nblk[1] * h->bg.nblk[2] + 2, false, h->stream));
  return 0;
}

extern "C" int mmd_comm_borders(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  h->nghost = 0;
  h->ghosts_uploaded = false;
  h->nghost_dev = nullptr;
  h->dh.ready = false; h->dh.pending = false;
  {
    const bool defer = h->in_reneighbor && h->opt_tiles && h->neigh_ready;
    h->borders_direct_pending = false;
    h->sendlists_stale = false;
    h->dh.gmap_live = false; h->dh.x_unpack_pending = false;
    int rc = borders_direct(h, defer);             // several ranks: the 26 lists in one exchange (they are the plan of the per-step halo too)
    if(rc < 0) return rc;
    const bool direct = rc >= 1;
    if(!direct) rc = borders_device_resident(h, defer);
    if(rc < 0) return rc;
    if(rc == 1) MMD_TRY(mmd_set_dummy(h));
    if(rc >= 1) {
      // several ranks: the lists of the direct per-step halo; their lengths arrive with the next host synchronisation (deferred: the neighbor build's)
      if(!direct) MMD_TRY(dh_enqueue(h));
      if(rc == 1 && h->dh.pending) { HIP_TRY(mmd_stream_sync(h)); MMD_TRY(dh_finish(h)); }
      h->neigh_nlocal = 0;
      h->tiles_ready = false;
      h->cand_src_ready = false;
      return 0;
    }
    h->nghost = 0;
  }
  return borders_general(h);
}

// deferred one-rank borders: the bst copy has arrived in h_flags_big (the caller synchronised). 1 = host state filled in,
// 0 = the estimates were too small: borders redone swap by swap (the caller must bin and build again), < 0 error
int mmd_borders_deferred_finish(mmd_handle* h)
{
  h->nghost_dev = nullptr;
  const int rc = borders_fast_finish(h);
  if(rc == 1) MMD_TRY(dh_finish(h));
  if(rc != 0) return rc;
  h->nghost = 0;
  h->dh.ready = false; h->dh.pending = false;
  const int rg = borders_general(h);
  return rg < 0 ? rg : 0;
}
// the same when nobody else is about to synchronise: fetch bst first
int mmd_borders_deferred_resolve(mmd_handle* h)
{
  if(!h->nghost_dev) return 1;
  HIP_TRY(hipMemcpyAsync(h->h_flags_big, h->bstate.p, 40 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  const int rc = mmd_borders_deferred_finish(h);
  if(rc == 1) MMD_TRY(mmd_set_dummy(h));
  return rc < 0 ? rc : 1;
}

static int borders_general(mmd_handle* h)
{
  int iswap = 0;
  h->sendlists_stale = false; h->borders_direct_pending = false;
  MMD_TRY(h->ghost_image.ensure(1024, false, h->stream));
  MMD_TRY(h->ghost_root.ensure(1024, false, h->stream));
  h->ghost_chain_ok = true;                // stays true while every swap is a self swap
  // the image code of a ghost holds at most +-2 box lengths per dimension (image_add clamps): a box thinner than cutneigh/2
  // (need > 2, possible with LAMMPS data files) must not rebuild its ghosts from root + image code
  for(int d = 0; d < 3; d++) if(h->need[d] > 2) h->ghost_chain_ok = false;
  // one pass over the owned atoms keeps only those inside some send slab (~12% at -s 80); the per-swap
  // selections then scan that short list + the ghosts instead of every owned atom six times
  int nb = -1;
  if(h->swaps.size() <= 6 && (h->nlocal > 4096 || h->nprocs > 1 || h->opt_force_transport)) {
    AnySlabPred ap;
    ap.x = h->x.p; ap.n = (int)h->swaps.size();
    for(int q = 0; q < ap.n; q++) { ap.lo[q] = h->swaps[q].slablo; ap.hi[q] = h->swaps[q].slabhi; ap.dim[q] = h->swaps[q].dim; }
    MMD_TRY(compact(h, ap, 0, h->nlocal, h->bnd_list, &nb));
    h->prev_nb = nb;
  }
  for(int d = 0; d < 3; d++) {
    int nfirst = 0, nlast = 0;
    for(int layer = 0; layer < h->need[d]; layer++, iswap += 2) {
      // the two swaps of this ghost layer scan the same atoms [nfirst, nlast) (ref/comm.cpp:766-770: nlast is frozen at the
      // even swap), so neither depends on the other's ghosts
      nfirst = nlast; nlast = h->nlocal + h->nghost;
      Swap* sw[2] = {&h->swaps[iswap], &h->swaps[iswap + 1]};
      int nsend[2] = {0, 0};
      for(int q = 0; q < 2; q++) {
        Swap& s = *sw[q];
        if(nb >= 0 && nfirst == 0)
          MMD_TRY(compact(h, BndSlabPred{h->x.p, h->bnd_list.p, nb, h->nlocal, d, s.slablo, s.slabhi}, 0, nb + (nlast - h->nlocal), s.sendlist, &nsend[q]));
        else
          MMD_TRY(compact(h, SlabPred{h->x.p, d, s.slablo, s.slabhi}, nfirst, nlast - nfirst, s.sendlist, &nsend[q]));
      }
      const bool self0 = sw[0]->sendproc == h->me && !h->opt_force_transport, self1 = sw[1]->sendproc == h->me && !h->opt_force_transport;
      if(self0 && self1) {
        for(int q = 0; q < 2; q++) {
          Swap& s = *sw[q];
          const real sx = s.pbc[0] * h->prd[0], sy = s.pbc[1] * h->prd[1], sz = s.pbc[2] * h->prd[2];
          const int nall = h->nlocal + h->nghost, nrecv = nsend[q];
          MMD_TRY(mmd_ensure_atoms(h, nall + nrecv + 1, true));
          MMD_TRY(h->ghost_image.ensure((size_t)h->nghost + nrecv + 8, true, h->stream, (size_t)h->nghost));
          MMD_TRY(h->ghost_root.ensure((size_t)h->nghost + nrecv + 8, true, h->stream, (size_t)h->nghost));
          if(nsend[q])                                   // (the pack kernel of a self swap writes the ghost types too)
            hipLaunchKernelGGL(k_pack_border, dim3(div_up(nsend[q], 256)), dim3(256), 0, h->stream, h->x.p, h->ghost_image.p, h->ghost_root.p,
                               h->nlocal, s.sendlist.p, nsend[q], sx, sy, sz, s.pbc_any, s.pbc[0], s.pbc[1], s.pbc[2], h->x.p + nall,
                               h->ghost_image.p + h->nghost, h->ghost_root.p + h->nghost, h->type.p + nall);
          HIP_TRY(hipGetLastError());
          s.sendnum = nsend[q]; s.recvnum = nrecv; s.firstrecv = nall;
          h->nghost += nrecv;
        }
        continue;
      }
      // ---- remote partners (a dimension is either all-self or all-remote): message = n real4 followed by n image codes
      h->ghost_chain_ok = false;
      const size_t rec = sizeof(real4) + sizeof(int);
      const size_t bytes_s[2] = {(size_t)nsend[0] * rec, (size_t)nsend[1] * rec};
      const size_t off_s1 = (bytes_s[0] + 63) & ~(size_t)63;                  // second message starts 64-byte aligned
      MMD_TRY(h->buf_send.ensure((off_s1 + bytes_s[1]) / sizeof(real) + 16, false, h->stream));
      unsigned char* sbase = (unsigned char*)h->buf_send.p;
      const void* dsend[2] = {sbase, sbase + off_s1};
      for(int q = 0; q < 2; q++) {
        Swap& s = *sw[q];
        const real sx = s.pbc[0] * h->prd[0], sy = s.pbc[1] * h->prd[1], sz = s.pbc[2] * h->prd[2];
        real4* dst = (real4*)dsend[q];
        if(nsend[q])
          hipLaunchKernelGGL(k_pack_border, dim3(div_up(nsend[q], 256)), dim3(256), 0, h->stream, h->x.p, h->ghost_image.p, h->ghost_root.p,
                             h->nlocal, s.sendlist.p, nsend[q], sx, sy, sz, s.pbc_any, s.pbc[0], s.pbc[1], s.pbc[2], dst, (int*)(dst + nsend[q]),
                             (int*)nullptr, (int*)nullptr);
      }
      HIP_TRY(hipGetLastError());
      const int dest[2] = {sw[0]->sendproc, sw[1]->sendproc}, src[2] = {sw[0]->recvproc, sw[1]->recvproc};
      int nrecv[2] = {0, 0};
      MMD_TRY(mmd_transport_sendrecv_counts_pair(h, nsend, dest, nrecv, src));
      const size_t bytes_r[2] = {(size_t)nrecv[0] * rec, (size_t)nrecv[1] * rec};
      const size_t off_r1 = (bytes_r[0] + 63) & ~(size_t)63;
      MMD_TRY(h->buf_recv.ensure((off_r1 + bytes_r[1]) / sizeof(real) + 16, false, h->stream));
      unsigned char* rbase = (unsigned char*)h->buf_recv.p;
      void* drecv[2] = {rbase, rbase + off_r1};
      MMD_TRY(mmd_transport_sendrecv_pair(h, dsend, bytes_s, dest, drecv, bytes_r, src));
      const int nall0 = h->nlocal + h->nghost;
      MMD_TRY(mmd_ensure_atoms(h, nall0 + nrecv[0] + nrecv[1] + 1, true));
      MMD_TRY(h->ghost_image.ensure((size_t)h->nghost + nrecv[0] + nrecv[1] + 8, true, h->stream, (size_t)h->nghost));
      for(int q = 0; q < 2; q++) {
        Swap& s = *sw[q];
        const int nall = h->nlocal + h->nghost;
        if(nrecv[q]) {
          HIP_TRY(hipMemcpyAsync(h->x.p + nall, drecv[q], (size_t)nrecv[q] * sizeof(real4), hipMemcpyDeviceToDevice, h->stream));
          HIP_TRY(hipMemcpyAsync(h->ghost_image.p + h->nghost, (real4*)drecv[q] + nrecv[q], (size_t)nrecv[q] * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
          hipLaunchKernelGGL(k_ghost_types, dim3(div_up(nrecv[q], 256)), dim3(256), 0, h->stream, h->x.p, nall, nrecv[q], h->type.p);
        }
        HIP_TRY(hipGetLastError());
        s.sendnum = nsend[q]; s.recvnum = nrecv[q]; s.firstrecv = nall;
        h->nghost += nrecv[q];
      }
    }
  }
  MMD_TRY(mmd_set_dummy(h));
  h->prev_nghost = h->nghost;
  h->borders_general_done = true;
  h->borders_general_runs++;
  MMD_TRY(dh_enqueue(h));
  if(h->dh.pending) { HIP_TRY(mmd_stream_sync(h)); MMD_TRY(dh_finish(h)); }
  h->neigh_nlocal = 0;                 // any existing neighbor list is stale now
  h->tiles_ready = false;
  h->cand_src_ready = false;
  return 0;
}

// interior tiles (no ghost among their candidates) first, boundary tiles after: the former can run while the halo
// of this step is still in flight on the communication stream
int mmd_order_tiles(mmd_handle* h)
{
  if(h->ntiles_interior >= 0) return 0;
  const int nt = h->ntiles;
  MMD_TRY(h->tile_order.ensure((size_t)nt + 8, false, h->stream));
  DevArr<int> part;
  int n_int = 0, n_bnd = 0;
  MMD_TRY(compact(h, TileFlagPred{h->tile_ghost.p, 0}, 0, nt, part, &n_int));
  if(n_int) HIP_TRY(hipMemcpyAsync(h->tile_order.p, part.p, (size_t)n_int * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  MMD_TRY(compact(h, TileFlagPred{h->tile_ghost.p, 1}, 0, nt, part, &n_bnd));
  if(n_bnd) HIP_TRY(hipMemcpyAsync(h->tile_order.p + n_int, part.p, (size_t)n_bnd * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  part.release();
  if(n_int + n_bnd != nt) { mmd_set_error("mmd_order_tiles: lost tiles (%d + %d != %d)", n_int, n_bnd, nt); return -1; }
  h->ntiles_interior = n_int;
  return 0;
}
