// minimd_amd/csrc/host.cpp — host-side setup shared by the `miniMD` executable and the Python mirror:
// input deck (ref/input.cpp), lattice + velocities (ref/setup.cpp), EAM funcfl -> splines
// (ref/force_eam.cpp:505-793). Pure host C++ (no HIP calls): usable without a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "mmd_internal.hpp"

// ---------------------------------------------------------------------------------------------------
// input() — ref/input.cpp:48-187.  Fixed positional format: 14 lines, the first two are ignored,
// then one record per line, only the leading token(s) are read.
// ---------------------------------------------------------------------------------------------------
static bool to_real(const std::string& tok, mmd_float* out)
{
  char* end = nullptr;
#if MMD_PRECISION == 1
  *out = strtof(tok.c_str(), &end);
#else
  *out = strtod(tok.c_str(), &end);
#endif
  return end != tok.c_str();
}

extern "C" int mmd_input_read(mmd_input* in, const char* filename)
{
  if(!in || !filename) { mmd_set_error("mmd_input_read: bad arguments"); return 1; }
  std::ifstream fs(filename);
  if(!fs) { mmd_set_error("ERROR: Cannot open %s", filename); return 1; }
  std::vector<std::vector<std::string>> rec;
  std::string line;
  while(rec.size() < 14 && std::getline(fs, line)) {
    std::istringstream is(line);
    std::vector<std::string> toks;
    std::string t;
    while(is >> t) toks.push_back(t);
    rec.push_back(toks);
  }
  if(rec.size() < 14) { mmd_set_error("ERROR: %s has fewer than 14 lines", filename); return 1; }
  auto tok = [&](int ln, size_t k) -> std::string { return rec[ln].size() > k ? rec[ln][k] : std::string(); };
  memset(in, 0, sizeof(*in));
  if(tok(2, 0) == "lj") in->units = 0;
  else if(tok(2, 0) == "metal") in->units = 1;
  else { mmd_set_error("Unknown units option in file at line 3 ('%s'). Expecting either 'lj' or 'metal'.", tok(2, 0).c_str()); return 1; }
  if(tok(3, 0) == "none") in->has_datafile = 0;
  else { in->has_datafile = 1; strncpy(in->datafile, tok(3, 0).c_str(), sizeof(in->datafile) - 1); }
  if(tok(4, 0) == "lj") in->forcetype = 0;
  else if(tok(4, 0) == "eam") in->forcetype = 1;
  else { mmd_set_error("Unknown forcetype option in file at line 5 ('%s'). Expecting either 'lj' or 'eam'.", tok(4, 0).c_str()); return 1; }
  mmd_float skin = 0;
  bool ok = to_real(tok(5, 0), &in->epsilon) && to_real(tok(5, 1), &in->sigma);
  in->nx = atoi(tok(6, 0).c_str()); in->ny = atoi(tok(6, 1).c_str()); in->nz = atoi(tok(6, 2).c_str());
  in->ntimes = atoi(tok(7, 0).c_str());
  ok = ok && to_real(tok(8, 0), &in->dt) && to_real(tok(9, 0), &in->t_request) && to_real(tok(10, 0), &in->rho);
  in->neigh_every = atoi(tok(11, 0).c_str());
  ok = ok && to_real(tok(12, 0), &in->force_cut) && to_real(tok(12, 1), &skin);
  in->thermo_nstat = atoi(tok(13, 0).c_str());
  if(!ok) { mmd_set_error("ERROR: %s: malformed numeric field", filename); return 1; }
  in->neigh_cut = skin + in->force_cut;          // the deck stores the skin (ref/input.cpp:183)
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// create_box / create_atoms — ref/setup.cpp:305-450
// ---------------------------------------------------------------------------------------------------
extern "C" int mmd_create_box(int nx, int ny, int nz, double rho, mmd_float prd[3])
{
  const double lattice = pow(4.0 / rho, 1.0 / 3.0);
  prd[0] = nx * lattice; prd[1] = ny * lattice; prd[2] = nz * lattice;
  return 0;
}

namespace {
// Park–Miller "minimal standard" generator in Schrage's overflow-free form (ref/setup.cpp:505-517)
struct ParkMiller {
  int state;
  double next()
  {
    const int q = state / 127773;
    state = 16807 * (state - q * 127773) - 2836 * q;
    if(state < 0) state += 2147483647;
    return (1.0 / 2147483647) * state;
  }
};
// glibc rand() after srand(5413) (ref/ljs.cpp:110, ref/atom.cpp:97), kept per call so that ranks
// sharing a process stay independent
struct TypeStream {
  struct random_data rd;
  char state[128];
  TypeStream()
  {
    memset(&rd, 0, sizeof(rd));
    initstate_r(1, state, sizeof(state), &rd);
    srandom_r(5413, &rd);
  }
  int next(int ntypes) { int32_t v; random_r(&rd, &v); return v % ntypes; }
};
}  // namespace

extern "C" int mmd_create_atoms(int nx, int ny, int nz, double rho, const mmd_float lo[3], const mmd_float hi[3], int ntypes,
                                mmd_float* x, mmd_float* v, int* type, int* tag, int* nlocal)
{
  if(!nlocal || ntypes < 1) { mmd_set_error("mmd_create_atoms: bad arguments"); return -1; }
  const double alat = pow(4.0 / rho, 1.0 / 3.0), half = 0.5 * alat;
  int lo_i[3], hi_i[3];
  const int nn[3] = {nx, ny, nz};
  for(int d = 0; d < 3; d++) {
    lo_i[d] = static_cast<int>(lo[d] / half - 1);
    hi_i[d] = static_cast<int>(hi[d] / half + 1);
    if(lo_i[d] < 0) lo_i[d] = 0;
    if(hi_i[d] > 2 * nn[d] - 1) hi_i[d] = 2 * nn[d] - 1;
  }
  const bool fill = x != nullptr;
  TypeStream types;
  int count = 0;
  // half-lattice sites are visited tile by tile (8x8x8 sites, x fastest both across and inside tiles)
  // so that creation order is spatially coherent — ref/setup.cpp:355-422
  const int TILE = 8;
  for(int tz = 0; tz * TILE <= hi_i[2]; tz++)
    for(int ty = 0; ty * TILE <= hi_i[1]; ty++)
      for(int tx = 0; tx * TILE <= hi_i[0]; tx++)
        for(int sz = 0; sz < TILE; sz++)
          for(int sy = 0; sy < TILE; sy++)
            for(int sx = 0; sx < TILE; sx++) {
              const int i = tx * TILE + sx, j = ty * TILE + sy, k = tz * TILE + sz;
              if((i + j + k) & 1) continue;                       // FCC: even parity sites only
              if(i < lo_i[0] || i > hi_i[0] || j < lo_i[1] || j > hi_i[1] || k < lo_i[2] || k > hi_i[2]) continue;
              const double px = half * i, py = half * j, pz = half * k;
              if(!(px >= lo[0] && px < hi[0] && py >= lo[1] && py < hi[1] && pz >= lo[2] && pz < hi[2])) continue;
              if(fill) {
                const int site = k * (2 * ny) * (2 * nx) + j * (2 * nx) + i + 1;   // global id seeds the velocity
                ParkMiller rng{site};
                double vel[3];
                for(int c = 0; c < 3; c++) {
                  for(int warm = 0; warm < 5; warm++) rng.next();
                  vel[c] = rng.next();
                }
                x[3 * count + 0] = px; x[3 * count + 1] = py; x[3 * count + 2] = pz;
                v[3 * count + 0] = vel[0]; v[3 * count + 1] = vel[1]; v[3 * count + 2] = vel[2];
                type[count] = types.next(ntypes);
                if(tag) tag[count] = site;
              }
              count++;
            }
  *nlocal = count;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// LAMMPS data file — read_lammps_data and helpers, ref/setup.cpp:55-301 (host part: parsing; the caller
// does comm/neighbor/thermo setup and keeps the atoms of its sub-box)
// ---------------------------------------------------------------------------------------------------
namespace {
struct LammpsData {
  FILE* fp = nullptr;
  std::string keyword;                      // current section name, empty at end of file
  static bool blank(const char* l) { return strspn(l, " \t\n\r") == strlen(l); }
  // next section keyword (ref :55-93): the first non-blank line (optionally the one already in `cur`), trimmed;
  // the line after it is consumed too
  void next_keyword(char* cur, bool have_cur)
  {
    char buf[1024];
    bool eof = false;
    if(!have_cur && !fgets(cur, 1024, fp)) eof = true;
    while(!eof && blank(cur)) if(!fgets(cur, 1024, fp)) eof = true;
    if(!eof && !fgets(buf, 1024, fp)) eof = true;
    if(eof) { keyword.clear(); return; }
    const size_t a = strspn(cur, " \t\n\r");
    size_t b = strlen(cur);
    while(b > a && strchr(" \t\n\r", cur[b - 1])) b--;
    keyword.assign(cur + a, b - a);
  }
};
}  // namespace

extern "C" int mmd_lammps_data_read(const char* file, int* natoms, mmd_float prd[3], mmd_float* mass, mmd_float* x, mmd_float* v)
{
  if(!file || !natoms || !prd) { mmd_set_error("mmd_lammps_data_read: bad arguments"); return -1; }
  LammpsData d;
  d.fp = fopen(file, "r");
  if(!d.fp) { mmd_set_error("Cannot open file %s", file); return -1; }
  char line[1024];
  // ---- header (ref :95-163): line 1 is a title; then "<n> atoms", "<n> atom types", "<lo> <hi> xlo xhi" ...
  // ('#' starts a comment, blank lines are skipped) until the first line that is none of these
  *natoms = 0;
  prd[0] = prd[1] = prd[2] = 0;
  bool have_line = false;
  if(!fgets(line, sizeof(line), d.fp)) line[0] = '\0';
  while(true) {
    if(!fgets(line, sizeof(line), d.fp)) { line[0] = '\0'; break; }
    if(char* c = strchr(line, '#')) *c = '\0';
    if(LammpsData::blank(line)) continue;
    double lo = 0, hi = 0;
    int ntypes_file = 0;
    if(strstr(line, "atoms")) sscanf(line, "%i", natoms);
    else if(strstr(line, "atom types")) sscanf(line, "%i", &ntypes_file);     // read and ignored, like the reference
    else if(strstr(line, "xlo xhi")) { sscanf(line, "%lg %lg", &lo, &hi); prd[0] = hi - lo; }
    else if(strstr(line, "ylo yhi")) { sscanf(line, "%lg %lg", &lo, &hi); prd[1] = hi - lo; }
    else if(strstr(line, "zlo zhi")) { sscanf(line, "%lg %lg", &lo, &hi); prd[2] = hi - lo; }
    else { have_line = true; break; }
  }
  if(*natoms <= 0 || !(prd[0] > 0 && prd[1] > 0 && prd[2] > 0)) {
    fclose(d.fp);
    mmd_set_error("%s: data file header needs '<n> atoms' and the xlo xhi / ylo yhi / zlo zhi lines", file);
    return -1;
  }
  if(!have_line) { fclose(d.fp); mmd_set_error("%s: no Atoms section", file); return -1; }
  if(mass) *mass = -1;                       // stays -1 when the file has no Masses section
  const bool fill = x && v;
  if(fill) for(size_t i = 0; i < (size_t)3 * *natoms; i++) { x[i] = 0; v[i] = 0; }
  d.next_keyword(line, true);
  bool atoms_seen = false;
  int rc = 0;
  while(!d.keyword.empty() && rc == 0) {
    if(d.keyword == "Atoms" || d.keyword == "Velocities") {          // ref :165-216: <id> [<type>] <a> <b> <c>, 1-based ids
      const bool pos = d.keyword == "Atoms";
      if(!pos && !atoms_seen) { mmd_set_error("%s: Must read Atoms before Velocities", file); rc = -1; break; }
      for(int n = 0; n < *natoms; n++) {
        if(!fgets(line, sizeof(line), d.fp)) { mmd_set_error("%s: unexpected end of file in section %s", file, d.keyword.c_str()); rc = -1; break; }
        int id = 0, type = 0;
        double a = 0, b = 0, c = 0;
        const int got = pos ? sscanf(line, "%i %i %lg %lg %lg", &id, &type, &a, &b, &c) : sscanf(line, "%i %lg %lg %lg", &id, &a, &b, &c);
        if(got != (pos ? 5 : 4) || id < 1 || id > *natoms) { mmd_set_error("%s: bad line in section %s: %s", file, d.keyword.c_str(), line); rc = -1; break; }
        if(fill) { mmd_float* q = (pos ? x : v) + 3 * (size_t)(id - 1); q[0] = a; q[1] = b; q[2] = c; }
      }
      atoms_seen = atoms_seen || pos;
    } else if(d.keyword == "Masses") {                                 // ref :267-276: one line "<type> <mass>"
      double m = 0; int t = 0;
      if(fgets(line, sizeof(line), d.fp) && sscanf(line, "%i %lg", &t, &m) == 2 && mass) *mass = (mmd_float)m;
    } else {
      mmd_set_error("Unknown identifier in data file: %s", d.keyword.c_str());
      rc = -1;
      break;
    }
    if(rc == 0) d.next_keyword(line, false);
  }
  fclose(d.fp);
  if(rc == 0 && !atoms_seen) { mmd_set_error("%s: no Atoms section", file); rc = -1; }
  return rc;
}

// atoms of the sub-box [lo,hi) in file order (ref/setup.cpp:281-286, Atom::addatom ref/atom.cpp:86-100); tag = file id.
// Two-call protocol like mmd_create_atoms.
extern "C" int mmd_lammps_data_select(int natoms, const mmd_float* x_all, const mmd_float* v_all, const mmd_float lo[3], const mmd_float hi[3],
                                      int ntypes, mmd_float* x, mmd_float* v, int* type, int* tag, int* nlocal)
{
  if(!x_all || !v_all || !nlocal || ntypes < 1) { mmd_set_error("mmd_lammps_data_select: bad arguments"); return -1; }
  TypeStream types;
  int count = 0;
  for(int i = 0; i < natoms; i++) {
    const mmd_float* p = x_all + 3 * (size_t)i;
    if(!(p[0] >= lo[0] && p[0] < hi[0] && p[1] >= lo[1] && p[1] < hi[1] && p[2] >= lo[2] && p[2] < hi[2])) continue;
    if(x) {
      for(int d = 0; d < 3; d++) { x[3 * (size_t)count + d] = p[d]; v[3 * (size_t)count + d] = v_all[3 * (size_t)i + d]; }
      type[count] = types.next(ntypes);
      if(tag) tag[count] = i + 1;
    }
    count++;
  }
  *nlocal = count;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// EAM: DYNAMO funcfl file -> uniform-grid arrays -> 7-coefficient splines (ref/force_eam.cpp:505-793)
// ---------------------------------------------------------------------------------------------------
namespace {
struct Funcfl {
  int nrho = 0, nr = 0;
  double drho = 0, dr = 0, cut = 0, mass = 0;
  std::vector<mmd_float> frho, rhor, zr;     // 1-based like the reference (:575-579)
};

bool read_values(std::istream& is, int n, std::vector<mmd_float>& out)
{
  out.assign(n + 1, 0);
  for(int i = 1; i <= n; i++) {
    std::string t;
    if(!(is >> t)) return false;
    out[i] = atof(t.c_str());
  }
  return true;
}

// 4-point Lagrange interpolation of a 1-based table onto abscissa r (ref/force_eam.cpp:630-646)
double lagrange4(const std::vector<mmd_float>& tab, int n, double delta, double r)
{
  const double sixth = 1.0 / 6.0;
  double p = r / delta + 1.0;
  int k = static_cast<int>(p);
  if(k > n - 2) k = n - 2;
  if(k < 2) k = 2;
  p -= k;
  if(p > 2.0) p = 2.0;
  const double cof1 = -sixth * p * (p - 1.0) * (p - 2.0);
  const double cof2 = 0.5 * (p * p - 1.0) * (p - 2.0);
  const double cof3 = -0.5 * p * (p + 1.0) * (p - 2.0);
  const double cof4 = sixth * p * (p * p - 1.0);
  return cof1 * tab[k - 1] + cof2 * tab[k] + cof3 * tab[k + 1] + cof4 * tab[k + 2];
}

// cubic spline in the reference's 7-slot form: [6]=value, [5..3]=cubic coeffs, [2..0]=derivative coeffs
void spline7(int n, mmd_float delta, const std::vector<mmd_float>& f, mmd_float* s)
{
  auto S = [&](int m, int c) -> mmd_float& { return s[m * 7 + c]; };
  for(int m = 1; m <= n; m++) S(m, 6) = f[m];
  S(1, 5) = S(2, 6) - S(1, 6);
  S(2, 5) = 0.5 * (S(3, 6) - S(1, 6));
  S(n - 1, 5) = 0.5 * (S(n, 6) - S(n - 2, 6));
  S(n, 5) = S(n, 6) - S(n - 1, 6);
  for(int m = 3; m <= n - 2; m++) S(m, 5) = ((S(m - 2, 6) - S(m + 2, 6)) + 8.0 * (S(m + 1, 6) - S(m - 1, 6))) / 12.0;
  for(int m = 1; m <= n - 1; m++) {
    S(m, 4) = 3.0 * (S(m + 1, 6) - S(m, 6)) - 2.0 * S(m, 5) - S(m + 1, 5);
    S(m, 3) = S(m, 5) + S(m + 1, 5) - 2.0 * (S(m + 1, 6) - S(m, 6));
  }
  S(n, 4) = 0.0;
  S(n, 3) = 0.0;
  for(int m = 1; m <= n; m++) {
    S(m, 2) = S(m, 5) / delta;
    S(m, 1) = 2.0 * S(m, 4) / delta;
    S(m, 0) = 3.0 * S(m, 3) / delta;
  }
}
}  // namespace

extern "C" int mmd_eam_tables_from_file(const char* filename, int ntypes, int* nr_out, int* nrho_out, int* nr_tot_out,
                                        int* nrho_tot_out, mmd_float* rdr, mmd_float* rdrho, mmd_float* cutmax, mmd_float* mass,
                                        mmd_float* rhor_spline, mmd_float* frho_spline, mmd_float* z2r_spline)
{
  if(!filename || ntypes < 1) { mmd_set_error("mmd_eam_tables_from_file: bad arguments"); return -1; }
  std::ifstream fs(filename);
  if(!fs) { mmd_set_error("Can't open EAM Potential file: %s", filename); return -1; }
  Funcfl fl;
  std::string line;
  std::getline(fs, line);                                  // comment line
  std::getline(fs, line);
  { int z; std::istringstream is(line); is >> z >> fl.mass; }
  std::getline(fs, line);
  { std::istringstream is(line); is >> fl.nrho >> fl.drho >> fl.nr >> fl.dr >> fl.cut; }
  if(fl.nrho < 5 || fl.nr < 5) { mmd_set_error("%s: not a funcfl EAM file", filename); return -1; }
  // file order: F(rho), Z(r), rho(r)
  if(!read_values(fs, fl.nrho, fl.frho) || !read_values(fs, fl.nr, fl.zr) || !read_values(fs, fl.nr, fl.rhor)) {
    mmd_set_error("%s: truncated EAM table", filename);
    return -1;
  }
  // file2array (:589-728): one file => the common grid is the file's own grid
  const mmd_float dr = fl.dr, drho = fl.drho;
  const double rmax = (fl.nr - 1) * fl.dr, rhomax = (fl.nrho - 1) * fl.drho;
  const int nr = static_cast<int>(rmax / dr + 0.5), nrho = static_cast<int>(rhomax / drho + 0.5);
  int nrho_tot = (nrho + 1) * 7 + 64, nr_tot = (nr + 1) * 7 + 64;      // array2spline (:737-740)
  nrho_tot -= nrho_tot % 64;
  nr_tot -= nr_tot % 64;
  if(nr_out) *nr_out = nr;
  if(nrho_out) *nrho_out = nrho;
  if(nr_tot_out) *nr_tot_out = nr_tot;
  if(nrho_tot_out) *nrho_tot_out = nrho_tot;
  if(rdr) *rdr = 1.0 / dr;
  if(rdrho) *rdrho = 1.0 / drho;
  if(cutmax) *cutmax = fl.cut;
  if(mass) *mass = fl.mass;
  if(!rhor_spline || !frho_spline || !z2r_spline) return 0;     // size query only

  std::vector<mmd_float> frho(nrho + 1), rhor(nr + 1), z2r(nr + 1);
  for(int m = 1; m <= nrho; m++) { const double r = (m - 1) * drho; frho[m] = lagrange4(fl.frho, fl.nrho, fl.drho, r); }
  for(int m = 1; m <= nr; m++) { const double r = (m - 1) * dr; rhor[m] = lagrange4(fl.rhor, fl.nr, fl.dr, r); }
  for(int m = 1; m <= nr; m++) {
    const double r = (m - 1) * dr;
    const double zri = lagrange4(fl.zr, fl.nr, fl.dr, r);
    const double zrj = lagrange4(fl.zr, fl.nr, fl.dr, r);
    z2r[m] = 27.2 * 0.529 * zri * zrj;
  }
  const int nt2 = ntypes * ntypes;
  memset(frho_spline, 0, sizeof(mmd_float) * (size_t)nt2 * nrho_tot);
  memset(rhor_spline, 0, sizeof(mmd_float) * (size_t)nt2 * nr_tot);
  memset(z2r_spline, 0, sizeof(mmd_float) * (size_t)nt2 * nr_tot);
  spline7(nrho, drho, frho, frho_spline);
  spline7(nr, dr, rhor, rhor_spline);
  spline7(nr, dr, z2r, z2r_spline);
  for(int t = 1; t < nt2; t++) {                                   // replicate per type pair (:753-760)
    memcpy(frho_spline + (size_t)t * nrho_tot, frho_spline, sizeof(mmd_float) * nrho_tot);
    memcpy(rhor_spline + (size_t)t * nr_tot, rhor_spline, sizeof(mmd_float) * nr_tot);
    memcpy(z2r_spline + (size_t)t * nr_tot, z2r_spline, sizeof(mmd_float) * nr_tot);
  }
  return 0;
}
