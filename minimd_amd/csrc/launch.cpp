// minimd_amd/csrc/launch.cpp — the process-level contract of a multi-rank run (host only, no HIP call in this file).
//
// The reference takes its rank and size from MPI (MPI_Comm_rank / MPI_Comm_size, ref/ljs.cpp:63-68) and its harness starts
// `${MPISTART} -np N ./miniMD ...` (ref/run_one_test:50). This executable links no MPI: the same launchers still work because each of them
// describes the rank to its children through the environment — read here (mmd_launch_env) — and the ranks find each other over TCP
// (mmd_mesh_*): rank 0 listens on MASTER_ADDR:MASTER_PORT+17 or, when the launcher exports neither, on 127.0.0.1 and a port derived from the
// launcher's job id, which every rank of the job sees alike. The mesh carries the set-up (who sits on which host with how many GPUs, the 128-byte
// RCCL id) and, when the ranks cannot have a GPU each, IS the transport: mmd_mesh_sendrecv / mmd_mesh_allreduce have the signatures of
// mmd_comm_set_host_transport's callbacks (MPI_Sendrecv / MPI_Allreduce(SUM) semantics of ref/comm.cpp:291-297, ref/thermo.cpp:131-133).
// That host-staged path is a DEBUG transport — every message crosses PCIe twice — for running the reference's np = 3 / 8 validation cases on one
// GPU; the production transport is RCCL inside comm.hip, and bench.py labels a run that did not use it `valid: false`.
#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmd.h"

void mmd_set_error(const char* fmt, ...);
double mmd_wall();

// ---------------------------------------------------------------------------------------------------
// who am I: torchrun | Open MPI (orterun / prrte) | MPICH, Intel MPI (hydra: PMI) | Slurm (srun)
// ---------------------------------------------------------------------------------------------------
static bool env_int(const char* name, int* out)
{
  const char* e = getenv(name);
  if(!e || !*e) return false;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if(end == e) return false;
  *out = (int)v;
  return true;
}

// what the last mmd_launch_env believed, in words: goes into the message of a rendezvous nobody else shows up for
static char g_launch_belief[256] = "no launcher variables: single rank";

extern "C" int mmd_launch_env(int* rank, int* nranks, int* local_rank, int* local_size, char* launcher, int launcher_len)
{
  struct Src { const char* name; const char* rank; const char* size; const char* lrank; const char* lsize; };
  // Slurm exports SLURM_PROCID=0 / SLURM_NTASKS=N to the batch shell of `sbatch -n N` as well: a bare ./miniMD started there is a singleton for MPI, and for
  // us — the Slurm row counts only inside a job STEP (srun: numeric SLURM_STEP_ID) and takes its size from the step (SLURM_STEP_NUM_TASKS)
  static const Src srcs[] = {
      {"torchrun", "RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"},
      {"openmpi", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_SIZE"},
      {"pmi", "PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", "MPI_LOCALNRANKS"},
      {"pmix", "PMIX_RANK", "PMIX_SIZE", "PMIX_LOCAL_RANK", "PMIX_LOCAL_SIZE"},
      {"slurm", "SLURM_PROCID", "SLURM_STEP_NUM_TASKS", "SLURM_LOCALID", "SLURM_STEP_TASKS_PER_NODE"},
  };
  int r = 0, n = 1, lr = -1, ls = -1;
  const char* how = "single";
  // explicit override: MMD_LAUNCHER=none (or MMD_NRANKS=1) = a singleton whatever the environment holds (a stale RANK / WORLD_SIZE, a batch shell);
  // MMD_LAUNCHER=<row name> = believe that launcher's variables only
  const char* want = getenv("MMD_LAUNCHER");
  int forced_n = 0;
  const bool single = (want && (!strcmp(want, "none") || !strcmp(want, "single"))) || (env_int("MMD_NRANKS", &forced_n) && forced_n == 1);
  snprintf(g_launch_belief, sizeof(g_launch_belief), single ? "MMD_LAUNCHER=none / MMD_NRANKS=1: single rank" : "no launcher variables: single rank");
  for(const Src& s : srcs) {
    if(single) break;
    if(want && *want && strcmp(want, s.name)) continue;
    int rr, nn, step;
    if(!strcmp(s.name, "slurm") && !env_int("SLURM_STEP_ID", &step)) continue;      // (not inside srun: the batch shell's variables are not a launch)
    if(env_int(s.rank, &rr) && env_int(s.size, &nn)) {
      r = rr; n = nn; how = s.name;
      if(!env_int(s.lrank, &lr)) lr = -1;
      if(!env_int(s.lsize, &ls)) ls = -1;
      snprintf(g_launch_belief, sizeof(g_launch_belief), "%s=%d and %s=%d (%s) — MMD_LAUNCHER=none runs a single rank regardless", s.rank, rr, s.size, nn, s.name);
      break;
    }
  }
  if(n < 1 || r < 0 || r >= n) { mmd_set_error("launcher environment (%s): rank %d of %d makes no sense", how, r, n); return -1; }
  if(lr < 0) lr = r;                    // (one node assumed when the launcher does not say)
  if(ls < 1) ls = n;
  if(rank) *rank = r;
  if(nranks) *nranks = n;
  if(local_rank) *local_rank = lr;
  if(local_size) *local_size = ls;
  if(launcher && launcher_len > 0) { strncpy(launcher, how, launcher_len - 1); launcher[launcher_len - 1] = 0; }
  return 0;
}

static unsigned hash_str(const char* s, unsigned h = 2166136261u)
{
  for(; s && *s; s++) { h ^= (unsigned char)*s; h *= 16777619u; }
  return h;
}

// where the ranks meet: MASTER_ADDR / MASTER_PORT when exported (torchrun, the tests), otherwise the loop-back address and a port every rank of THIS
// job derives alike: from the launcher's job id, else from the parent's pid (mpiexec's proxy, srun's step daemon, or the shell that started the ranks)
extern "C" int mmd_launch_rendezvous(char* addr, int addr_len, int* port)
{
  const char* a = getenv("MASTER_ADDR");
  if(!a || !*a) a = "127.0.0.1";
  if(addr && addr_len > 0) { strncpy(addr, a, addr_len - 1); addr[addr_len - 1] = 0; }
  int p = 0;
  if(!env_int("MASTER_PORT", &p) || p <= 0) {
    unsigned h = 0;
    const char* keys[] = {"SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid", "OMPI_MCA_orte_ess_jobid", "PMIX_NAMESPACE", "PMI_JOBID", "PBS_JOBID", "LSB_JOBID"};
    bool any = false;
    for(const char* k : keys) { const char* e = getenv(k); if(e && *e) { h = hash_str(e, h ? h : 2166136261u); any = true; } }
    if(!any) { char b[32]; snprintf(b, sizeof(b), "ppid%ld", (long)getppid()); h = hash_str(b); }
    p = 20000 + (int)(h % 9973u);      // (below the kernel's ephemeral range, +17 added by the mesh)
  }
  if(port) *port = p;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// TCP mesh
// ---------------------------------------------------------------------------------------------------
struct mmd_mesh {
  int rank = 0, nranks = 1;
  std::vector<int> fd;                 // fd[r]: stream to rank r (-1: myself)
  std::vector<double> red;
  long long bytes_sent = 0, messages = 0;
};

static const uint32_t MESH_MAGIC = 0x6d6d6468u;      // "mmdh"
struct MeshHello { uint32_t magic; int32_t rank, nranks, port; uint32_t nonce; };
// a number every rank of THIS job derives alike and another job on the same port almost surely does not: the launcher's job id (or, without one, the
// rendezvous address itself), mixed with the world size
static uint32_t job_nonce(const char* addr, int port, int nranks)
{
  unsigned h = 2166136261u;
  const char* keys[] = {"SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid", "OMPI_MCA_orte_ess_jobid", "PMIX_NAMESPACE", "PMI_JOBID", "PBS_JOBID", "LSB_JOBID", "TORCHELASTIC_RUN_ID"};
  for(const char* k : keys) { const char* e = getenv(k); if(e && *e) h = hash_str(e, h); }
  char b[96];
  snprintf(b, sizeof(b), "%s:%d/%d", addr && *addr ? addr : "127.0.0.1", port, nranks);
  return hash_str(b, h);
}
// an accepted stream gets a few seconds to say who it is: a stranger that connects and sends nothing (a port scanner, another job that derived the same
// port) is dropped instead of stalling the job's start-up
static void set_rcv_timeout(int fd, int seconds)
{
  timeval tv{seconds, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
}
struct MeshHdr { uint32_t magic, tag; uint64_t nbytes; };

static int set_nonblock(int fd, bool on)
{
  const int fl = fcntl(fd, F_GETFL, 0);
  return fcntl(fd, F_SETFL, on ? (fl | O_NONBLOCK) : (fl & ~O_NONBLOCK));
}
static void tune(int fd)
{
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  int sz = 4 << 20;
  setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sz, sizeof(sz));
  setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &sz, sizeof(sz));
}
static bool write_all(int fd, const void* buf, size_t n)
{
  const char* p = (const char*)buf;
  while(n) { const ssize_t w = send(fd, p, n, MSG_NOSIGNAL); if(w < 0) { if(errno == EINTR) continue; return false; } p += w; n -= (size_t)w; }
  return true;
}
static bool read_all(int fd, void* buf, size_t n)
{
  char* p = (char*)buf;
  while(n) { const ssize_t r = recv(fd, p, n, 0); if(r < 0) { if(errno == EINTR) continue; return false; } if(r == 0) return false; p += r; n -= (size_t)r; }
  return true;
}
static bool resolve(const char* host, int port, sockaddr_in* sa)
{
  memset(sa, 0, sizeof(*sa));
  sa->sin_family = AF_INET; sa->sin_port = htons((uint16_t)port);
  if(inet_pton(AF_INET, host, &sa->sin_addr) == 1) return true;
  hostent* he = gethostbyname(host);
  if(!he || he->h_addrtype != AF_INET) return false;
  memcpy(&sa->sin_addr, he->h_addr_list[0], sizeof(sa->sin_addr));
  return true;
}
static int connect_retry(const sockaddr_in& sa, double seconds)
{
  const double t0 = mmd_wall();
  for(;;) {
    const int s = socket(AF_INET, SOCK_STREAM, 0);
    if(s < 0) return -1;
    if(connect(s, (const sockaddr*)&sa, sizeof(sa)) == 0) { tune(s); return s; }
    close(s);
    if(mmd_wall() - t0 > seconds) return -1;
    usleep(20000);
  }
}

extern "C" int mmd_mesh_destroy(mmd_mesh* m)
{
  if(!m) return 0;
  for(int f : m->fd) if(f >= 0) close(f);
  delete m;
  return 0;
}

// every rank of the job calls this with the same addr / port; returns once every pair of ranks shares a stream
extern "C" int mmd_mesh_create(int rank, int nranks, const char* addr, int port, mmd_mesh** out)
{
  if(!out || nranks < 1 || rank < 0 || rank >= nranks) { mmd_set_error("mmd_mesh_create: bad arguments"); return -1; }
  *out = nullptr;
  mmd_mesh* m = new mmd_mesh();
  m->rank = rank; m->nranks = nranks;
  m->fd.assign(nranks, -1);
  if(nranks == 1) { *out = m; return 0; }
  double patience = 120.0;                // (MMD_MESH_PATIENCE seconds: how long a rank waits for the others)
  { const char* e = getenv("MMD_MESH_PATIENCE"); if(e && *e && atof(e) > 0) patience = atof(e); }
  const int mport = port + 17;
  const uint32_t nonce = job_nonce(addr, port, nranks);
  // my own listener (ephemeral port) for the ranks above me: opened on the interface my stream to rank 0 uses (below), not on every interface
  int ls = -1, my_port = 0;
  auto open_listener = [&](uint32_t local_ip) -> bool {
    ls = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in sa;
    memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_addr.s_addr = local_ip; sa.sin_port = 0;
    socklen_t sl = sizeof(sa);
    if(ls < 0 || bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(ls, nranks) != 0 || getsockname(ls, (sockaddr*)&sa, &sl) != 0) {
      mmd_set_error("mesh: rank %d cannot open a listening socket: %s", rank, strerror(errno));
      if(ls >= 0) close(ls);
      ls = -1;
      return false;
    }
    my_port = ntohs(sa.sin_port);
    return true;
  };
  std::vector<uint32_t> ip(nranks, 0);
  std::vector<int32_t> ports(nranks, 0);
  if(rank == 0) {
    const int s0 = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(s0, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in sa;
    memset(&sa, 0, sizeof(sa));
    // listen where the ranks were told to meet (MASTER_ADDR, else loop-back) — on every interface only when that address is not one of this host's
    if(!resolve(addr && *addr ? addr : "127.0.0.1", mport, &sa)) { memset(&sa, 0, sizeof(sa)); sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_ANY); sa.sin_port = htons((uint16_t)mport); }
    bool bound = s0 >= 0 && bind(s0, (sockaddr*)&sa, sizeof(sa)) == 0;
    if(!bound && s0 >= 0 && errno == EADDRNOTAVAIL) { sa.sin_addr.s_addr = htonl(INADDR_ANY); bound = bind(s0, (sockaddr*)&sa, sizeof(sa)) == 0; }
    if(!bound || listen(s0, nranks) != 0) {
      mmd_set_error("mesh: rank 0 cannot listen on port %d (%s) — export MASTER_PORT to choose another one", mport, strerror(errno));
      if(s0 >= 0) close(s0);
      mmd_mesh_destroy(m);
      return -1;
    }
    for(int k = 1; k < nranks; k++) {
      pollfd pf{s0, POLLIN, 0};
      if(poll(&pf, 1, (int)(patience * 1000)) <= 0) {
        mmd_set_error("mesh: rank 0 waited %g s on port %d and %d of %d ranks arrived — the size was taken from %s", patience, mport, k, nranks, g_launch_belief);
        close(s0); mmd_mesh_destroy(m); return -1;
      }
      sockaddr_in pa;
      socklen_t pl = sizeof(pa);
      const int cs = accept(s0, (sockaddr*)&pa, &pl);
      MeshHello hl;
      if(cs >= 0) set_rcv_timeout(cs, 5);
      if(cs < 0 || !read_all(cs, &hl, sizeof(hl)) || hl.magic != MESH_MAGIC || hl.nonce != nonce || hl.nranks != nranks || hl.rank <= 0 || hl.rank >= nranks || m->fd[hl.rank] >= 0) {
        // (a stranger on the port — another job that derived the same number, a port scanner: not one of mine)
        if(cs >= 0) close(cs);
        k--;
        continue;
      }
      set_rcv_timeout(cs, 0);
      tune(cs);
      m->fd[hl.rank] = cs;
      ip[hl.rank] = pa.sin_addr.s_addr;
      ports[hl.rank] = hl.port;
    }
    close(s0);
    for(int k = 1; k < nranks; k++)
      if(!write_all(m->fd[k], ip.data(), nranks * sizeof(uint32_t)) || !write_all(m->fd[k], ports.data(), nranks * sizeof(int32_t))) {
        mmd_set_error("mesh: rank 0 lost rank %d during the rendezvous", k); mmd_mesh_destroy(m); return -1;
      }
  } else {
    sockaddr_in sa;
    if(!resolve(addr && *addr ? addr : "127.0.0.1", mport, &sa)) { mmd_set_error("mesh: cannot resolve '%s'", addr); mmd_mesh_destroy(m); return -1; }
    const int s = connect_retry(sa, patience);
    if(s >= 0 && rank < nranks - 1) {
      sockaddr_in me_sa;
      socklen_t ml = sizeof(me_sa);
      if(getsockname(s, (sockaddr*)&me_sa, &ml) != 0) me_sa.sin_addr.s_addr = htonl(INADDR_ANY);
      if(!open_listener(me_sa.sin_addr.s_addr)) { close(s); mmd_mesh_destroy(m); return -1; }
    }
    MeshHello hl{MESH_MAGIC, rank, nranks, my_port, nonce};
    if(s < 0 || !write_all(s, &hl, sizeof(hl)) || !read_all(s, ip.data(), nranks * sizeof(uint32_t)) || !read_all(s, ports.data(), nranks * sizeof(int32_t))) {
      mmd_set_error("mesh: rank %d of %d could not reach rank 0 at %s:%d (export MASTER_ADDR / MASTER_PORT if the launcher does not) — rank and size were taken from %s", rank, nranks,
                    addr ? addr : "127.0.0.1", mport, g_launch_belief);
      if(s >= 0) close(s);
      if(ls >= 0) close(ls);
      mmd_mesh_destroy(m);
      return -1;
    }
    m->fd[0] = s;
    // streams to the ranks below me (their listeners), from the ranks above me (my listener)
    for(int j = 1; j < rank; j++) {
      sockaddr_in pj;
      memset(&pj, 0, sizeof(pj));
      pj.sin_family = AF_INET; pj.sin_addr.s_addr = ip[j]; pj.sin_port = htons((uint16_t)ports[j]);
      const int c = connect_retry(pj, patience);
      MeshHello h2{MESH_MAGIC, rank, nranks, 0, nonce};
      if(c < 0 || !write_all(c, &h2, sizeof(h2))) { mmd_set_error("mesh: rank %d could not reach rank %d", rank, j); if(c >= 0) close(c); if(ls >= 0) close(ls); mmd_mesh_destroy(m); return -1; }
      m->fd[j] = c;
    }
    for(int need = nranks - 1 - rank; need > 0;) {
      pollfd pf{ls, POLLIN, 0};
      if(poll(&pf, 1, (int)(patience * 1000)) <= 0) { mmd_set_error("mesh: rank %d waited in vain for %d higher ranks", rank, need); close(ls); mmd_mesh_destroy(m); return -1; }
      const int cs = accept(ls, nullptr, nullptr);
      MeshHello h2;
      if(cs >= 0) set_rcv_timeout(cs, 5);
      if(cs < 0 || !read_all(cs, &h2, sizeof(h2)) || h2.magic != MESH_MAGIC || h2.nonce != nonce || h2.nranks != nranks || h2.rank <= rank || h2.rank >= nranks || m->fd[h2.rank] >= 0) { if(cs >= 0) close(cs); continue; }
      set_rcv_timeout(cs, 0);
      tune(cs);
      m->fd[h2.rank] = cs;
      need--;
    }
    if(ls >= 0) close(ls);
  }
  for(int r = 0; r < nranks; r++) if(r != rank) set_nonblock(m->fd[r], true);
  *out = m;
  return 0;
}

// send ns bytes to `dest` while receiving up to nr_max bytes from `src` (either side may be empty: nothing travels then, like the gloo transport of
// the tests); returns the bytes received, < 0 on error. Both directions progress together (poll), so two ranks that send to each other never wait
// on each other's buffers. A message = 16-byte header {magic, tag, length} + payload: a pattern that has come apart is noticed, not mis-read.
static long long mesh_xfer(mmd_mesh* m, const void* sbuf, long long ns, int dest, void* rbuf, long long nr_max, int src, uint32_t tag)
{
  if(ns < 0 || nr_max < 0 || dest < 0 || dest >= m->nranks || src < 0 || src >= m->nranks) { mmd_set_error("mesh: bad sendrecv arguments"); return -1; }
  const bool do_s = ns > 0, do_r = nr_max > 0;
  if(dest == m->rank && src == m->rank) {
    if(!do_s || !do_r) return 0;
    if(ns > nr_max) { mmd_set_error("mesh: self message of %lld bytes into %lld", ns, nr_max); return -1; }
    memmove(rbuf, sbuf, (size_t)ns);
    return ns;
  }
  if((do_s && dest == m->rank) || (do_r && src == m->rank)) { mmd_set_error("mesh: a self send must be matched by a self receive"); return -1; }
  MeshHdr hs{MESH_MAGIC, tag, (uint64_t)ns}, hr{0, 0, 0};
  size_t s_hdr = do_s ? 0 : sizeof(hs), s_pay = 0, r_hdr = do_r ? 0 : sizeof(hr), r_pay = 0;
  uint64_t r_len = 0;
  bool r_done = !do_r, s_done = !do_s;
  m->bytes_sent += ns; m->messages += do_s ? 1 : 0;
  while(!r_done || !s_done) {
    pollfd pf[2];
    int np = 0, is = -1, ir = -1;
    if(!s_done) { is = np; pf[np++] = pollfd{m->fd[dest], POLLOUT, 0}; }
    if(!r_done) {
      if(!s_done && src == dest) { pf[is].events |= POLLIN; ir = is; }
      else { ir = np; pf[np++] = pollfd{m->fd[src], POLLIN, 0}; }
    }
    const int pr = poll(pf, np, 120000);
    if(pr < 0 && errno == EINTR) continue;
    if(pr <= 0) { mmd_set_error("mesh: rank %d stuck in sendrecv (to %d: %lld bytes, from %d: up to %lld)", m->rank, dest, ns, src, nr_max); return -1; }
    if(!s_done && (pf[is].revents & (POLLOUT | POLLERR | POLLHUP))) {
      const char* p; size_t left;
      if(s_hdr < sizeof(hs)) { p = (const char*)&hs + s_hdr; left = sizeof(hs) - s_hdr; }
      else { p = (const char*)sbuf + s_pay; left = (size_t)ns - s_pay; }
      const ssize_t w = send(m->fd[dest], p, left, MSG_NOSIGNAL);
      if(w < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) { mmd_set_error("mesh: send to rank %d failed: %s", dest, strerror(errno)); return -1; }
      if(w > 0) { if(s_hdr < sizeof(hs)) s_hdr += (size_t)w; else s_pay += (size_t)w; }
      if(s_hdr == sizeof(hs) && s_pay == (size_t)ns) s_done = true;
    }
    if(!r_done && (pf[ir].revents & (POLLIN | POLLERR | POLLHUP))) {
      char* p; size_t left;
      if(r_hdr < sizeof(hr)) { p = (char*)&hr + r_hdr; left = sizeof(hr) - r_hdr; }
      else { p = (char*)rbuf + r_pay; left = (size_t)r_len - r_pay; }
      const ssize_t g = left ? recv(m->fd[src], p, left, 0) : 0;
      if(g == 0 && left) { mmd_set_error("mesh: rank %d closed the stream", src); return -1; }
      if(g < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) { mmd_set_error("mesh: receive from rank %d failed: %s", src, strerror(errno)); return -1; }
      if(g > 0) {
        if(r_hdr < sizeof(hr)) {
          r_hdr += (size_t)g;
          if(r_hdr == sizeof(hr)) {
            if(hr.magic != MESH_MAGIC || hr.tag != tag) { mmd_set_error("mesh: rank %d expected a message of kind %u from rank %d and found kind %u — the ranks' message patterns differ", m->rank, tag, src, hr.tag); return -1; }
            if(hr.nbytes > (uint64_t)nr_max) { mmd_set_error("mesh: message of %llu bytes from rank %d does not fit %lld", (unsigned long long)hr.nbytes, src, nr_max); return -1; }
            r_len = hr.nbytes;
          }
        } else r_pay += (size_t)g;
      }
      if(r_hdr == sizeof(hr) && r_pay == (size_t)r_len) r_done = true;
    }
  }
  return do_r ? (long long)r_len : 0;
}

extern "C" long long mmd_mesh_sendrecv(void* ctx, const void* sendbuf, long long nsend, int dest, void* recvbuf, long long nrecv_max, int src)
{
  return mesh_xfer((mmd_mesh*)ctx, sendbuf, nsend, dest, recvbuf, nrecv_max, src, 1u);
}

// every rank's `nbytes` at all ranks, in rank order (set-up data: host names, device counts, the RCCL id)
extern "C" int mmd_mesh_allgather(mmd_mesh* m, const void* mine, int nbytes, void* all)
{
  if(!m || nbytes <= 0 || !mine || !all) { mmd_set_error("mmd_mesh_allgather: bad arguments"); return -1; }
  char* a = (char*)all;
  memmove(a + (size_t)m->rank * nbytes, mine, (size_t)nbytes);
  if(m->rank == 0) {
    for(int r = 1; r < m->nranks; r++) if(mesh_xfer(m, nullptr, 0, 0, a + (size_t)r * nbytes, nbytes, r, 2u) != nbytes) return -1;
    for(int r = 1; r < m->nranks; r++) if(mesh_xfer(m, a, (long long)nbytes * m->nranks, r, nullptr, 0, 0, 3u) < 0) return -1;
  } else {
    if(mesh_xfer(m, mine, nbytes, 0, nullptr, 0, 0, 2u) < 0) return -1;
    if(mesh_xfer(m, nullptr, 0, 0, a, (long long)nbytes * m->nranks, 0, 3u) != (long long)nbytes * m->nranks) return -1;
  }
  return 0;
}

// in-place sum over the ranks, added up in rank order on rank 0 and handed back: every rank gets the same bits
extern "C" int mmd_mesh_allreduce(void* ctx, double* vals, int n)
{
  mmd_mesh* m = (mmd_mesh*)ctx;
  if(!m || n < 0 || (n && !vals)) { mmd_set_error("mmd_mesh_allreduce: bad arguments"); return -1; }
  if(m->nranks == 1 || n == 0) return 0;
  const long long nb = (long long)n * (long long)sizeof(double);
  if(m->rank == 0) {
    if(m->red.size() < (size_t)n) m->red.resize(n);
    for(int r = 1; r < m->nranks; r++) {
      if(mesh_xfer(m, nullptr, 0, 0, m->red.data(), nb, r, 4u) != nb) return -1;
      for(int i = 0; i < n; i++) vals[i] += m->red[i];
    }
    for(int r = 1; r < m->nranks; r++) if(mesh_xfer(m, vals, nb, r, nullptr, 0, 0, 5u) < 0) return -1;
  } else {
    if(mesh_xfer(m, vals, nb, 0, nullptr, 0, 0, 4u) < 0) return -1;
    if(mesh_xfer(m, nullptr, 0, 0, vals, nb, 0, 5u) != nb) return -1;
  }
  return 0;
}

extern "C" int mmd_mesh_info(mmd_mesh* m, int* rank, int* nranks, long long* bytes_sent, long long* messages)
{
  if(!m) { mmd_set_error("mmd_mesh_info: null mesh"); return -1; }
  if(rank) *rank = m->rank;
  if(nranks) *nranks = m->nranks;
  if(bytes_sent) *bytes_sent = m->bytes_sent;
  if(messages) *messages = m->messages;
  return 0;
}
