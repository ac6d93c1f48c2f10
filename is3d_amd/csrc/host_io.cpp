// host_io.cpp -- readers and writers in the reference's text formats (host side, plain C++17).
//
// Each function states the reference routine whose observable behaviour it reproduces
// (paths relative to /root/reference).  Nothing here calls exit(): errors become IS3D_E* codes.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <exception>
#include <memory>
#include <algorithm>
#include <charconv>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/is3d_amd.h"
#include "errors.h"

namespace {

const double kHbarC = 0.197327053;  // src/cpp/iS3D.h:9

bool slurp(const char *path, std::string &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    out.swap(s);
    return true;
}

// arsenal.cpp:552-565: trim() removes EVERY blank and tab, not only the ends
std::string strip_blanks(const std::string &s)
{
    std::string t;
    for (char ch : s)
        if (ch != ' ' && ch != '\t') t.push_back(ch);
    return t;
}
std::string lower(std::string s)
{
    for (char &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}

}  // namespace

namespace is3d {
static thread_local std::string g_last_error;
int set_error(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
static std::atomic<int64_t> g_resource[2];
void count_resource(int what) { if (what >= 0 && what < 2) g_resource[what].fetch_add(1, std::memory_order_relaxed); }
}  // namespace is3d
#define io_fail is3d::set_error

extern "C" int is3d_resource_counters(int64_t *plans_created, int64_t *device_allocations)
{
    if (plans_created) *plans_created = is3d::g_resource[0].load(std::memory_order_relaxed);
    if (device_allocations) *device_allocations = is3d::g_resource[1].load(std::memory_order_relaxed);
    return IS3D_OK;
}

extern "C" const char *is3d_last_error(void) { return is3d::g_last_error.c_str(); }

// ---------------------------------------------------------------------------------------------
// ParameterReader::readFromFile / getVal  (src/cpp/ParameterReader.cpp:38-155)
//   per line: drop everything from '#', remove all blanks/tabs, split at the first '=',
//   name lower-cased, value = leading double of the right-hand side; later lines overwrite.
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_param_get(const char *path, const char *name, double *value)
{
    if (!path || !name || !value) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "parameter file %s does not exist", path);
    const std::string want = lower(strip_blanks(name));
    bool found = false;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t nl = text.find('\n', pos);
        std::string line = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? text.size() + 1 : nl + 1;
        if (strip_blanks(line).empty()) continue;
        line = line.substr(0, line.find('#'));
        if (strip_blanks(line).empty()) continue;
        size_t eq = line.find('=');
        if (eq == std::string::npos)
            return io_fail(IS3D_EINVAL, "%s: \"=\" symbol not found in assignment \"%s\"", path, line.c_str());
        std::string lhs = lower(strip_blanks(line.substr(0, eq)));
        std::string rhs = strip_blanks(line.substr(eq + 1));
        if (lhs == want) {
            *value = strtod(rhs.c_str(), nullptr);
            found = true;
        }
    }
    if (!found) return io_fail(IS3D_EINVAL, "parameter with name %s not found in %s", name, path);
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Table::loadTableFromFile -> readBlockData  (src/cpp/Table.cpp:179-195, src/cpp/arsenal.cpp:406-453)
//   column count = numbers on the first line; every '\n'-terminated line is a row; whatever follows
//   the last '\n' is never stored.
// ---------------------------------------------------------------------------------------------
// strtod with Clinger's exact fast path in front: a decimal with at most 15 significant digits... precisely, a mantissa
// w <= 2^53 and a decimal exponent |e10| <= 22 is w * 10^e10 or w / 10^-e10 with BOTH operands exact doubles, hence correctly
// rounded by one IEEE operation -- the same value strtod returns, at a tenth of its cost.  Hydro codes print 6-9 digits;
// anything longer (or hex, inf, nan, ...) goes to strtod.
static double fast_strtod(const char *p, char **endp)
{
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17,
                                   1e18, 1e19, 1e20, 1e21, 1e22};
    const char *s = p;
    while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r' || *s == '\v' || *s == '\f') s++;
    const char *q = s;
    bool neg = false;
    if (*q == '+' || *q == '-') { neg = (*q == '-'); q++; }
    uint64_t w = 0;
    int nd = 0, e10 = 0;
    bool any = false;
    while (*q >= '0' && *q <= '9') { if (nd < 19) { w = w * 10 + (uint64_t)(*q - '0'); if (w) nd++; } else return strtod(p, endp); q++; any = true; }
    if (*q == '.') {
        q++;
        while (*q >= '0' && *q <= '9') { if (nd < 19) { w = w * 10 + (uint64_t)(*q - '0'); if (w) nd++; e10--; } else return strtod(p, endp); q++; any = true; }
    }
    if (!any) return strtod(p, endp);          // inf, nan, garbage: strtod decides (and reports "no conversion")
    if (*q == 'e' || *q == 'E') {
        const char *r = q + 1;
        bool eneg = false;
        if (*r == '+' || *r == '-') { eneg = (*r == '-'); r++; }
        if (*r >= '0' && *r <= '9') {
            int ex = 0;
            while (*r >= '0' && *r <= '9') { if (ex < 10000) ex = ex * 10 + (*r - '0'); r++; }
            e10 += eneg ? -ex : ex;
            q = r;
        }
    } else if (*q == 'x' || *q == 'X' || *q == 'p' || *q == 'P') return strtod(p, endp);   // hex float
    if (w > (1ULL << 53) || e10 < -22 || e10 > 22) return strtod(p, endp);
    double x = (double)w;
    x = e10 < 0 ? x / p10[-e10] : x * p10[e10];
    *endp = const_cast<char *>(q);
    return neg ? -x : x;
}

// arsenal.cpp:378-393 stringToDoubles.  Numbers of the line [b, e) parsed in place (strtod would run over the newline: blanks are skipped by hand), the first
// `keep` of them stored; returns how many the line holds up to the first token that does not parse (stringToDoubles).
static int parse_line(const char *b, const char *e, double *out, int keep)
{
    int n = 0;
    const char *p = b;
    for (;;) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
        if (p >= e) break;
        char *q;
        const double x = fast_strtod(p, &q);
        if (q == p || q > e) break;
        if (n < keep) out[n] = x;
        n++;
        p = q;
    }
    return n;
}

// A 1e6-cell surface is ~0.5 GB of text: the rows are parsed by up to 16 threads (strtod is the whole cost), each writing
// its own rows of the table.  *exact (optional) = every row holds exactly n_cols numbers, i.e. the row-major table is the
// file's whitespace token stream (what the surface readers, which consume tokens like operator>>, rely on).
static int parse_table(const std::string &text, const char *path, int64_t *n_rows, int32_t *n_cols,
                       std::vector<double> *store, bool *exact = nullptr)
{
    const char *base = text.data();
    std::vector<size_t> nl;
    for (const char *p = base, *end = base + text.size(); p < end;) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!q) break;
        nl.push_back((size_t)(q - base));
        p = q + 1;
    }
    const int32_t ncol = parse_line(base, base + (nl.empty() ? text.size() : nl[0]), nullptr, 0);
    if (ncol == 0) return io_fail(IS3D_EIO, "%s: empty first row; no data read", path);
    const int64_t rows = (int64_t)nl.size();   // every '\n'-terminated line is a row
    if (store) store->assign((size_t)rows * ncol, 0.0);
    int nthreads = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), rows / 8192 + 1);
    std::vector<int64_t> bad_row((size_t)nthreads, -1);
    std::vector<int> bad_cnt((size_t)nthreads, 0), inexact((size_t)nthreads, 0);
    auto work = [&](int t) {
        const int64_t r0 = rows * t / nthreads, r1 = rows * (t + 1) / nthreads;
        for (int64_t r = r0; r < r1; r++) {
            const char *b = base + (r == 0 ? 0 : nl[(size_t)r - 1] + 1), *e = base + nl[(size_t)r];
            const int cnt = parse_line(b, e, store ? store->data() + (size_t)r * ncol : nullptr, store ? ncol : 0);
            if (cnt < ncol && bad_row[t] < 0) { bad_row[t] = r; bad_cnt[t] = cnt; }
            if (cnt != ncol) inexact[t] = 1;
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    bool ex = true;
    for (int t = 0; t < nthreads; t++) {
        if (bad_row[t] >= 0)   // threads own ascending row ranges: the first hit is the earliest bad row
            return io_fail(IS3D_EIO, "%s: row %lld has %d numbers, expected %d", path, (long long)bad_row[t] + 1, bad_cnt[t], ncol);
        if (inexact[t]) ex = false;
    }
    if (exact) *exact = ex;
    *n_rows = rows;
    *n_cols = ncol;
    return IS3D_OK;
}

extern "C" int is3d_table_read(const char *path, int64_t *n_rows, int32_t *n_cols, double *data, int64_t capacity)
{
    if (!path || !n_rows || !n_cols) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    std::vector<double> store;
    int rc = parse_table(text, path, n_rows, n_cols, data ? &store : nullptr);
    if (rc) return rc;
    if (data) {
        if ((int64_t)store.size() > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %lld < %zu", path, (long long)capacity, store.size());
        memcpy(data, store.data(), store.size() * sizeof(double));
    }
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// FO_data_reader::get_number_cells + read_surf_VH  (src/cpp/readindata.cpp:122-131, :320-468)
//   cell count = Table rows of the file; values = whitespace token stream, 20 (+1 muB, +4 nB,Vx,Vy,Vn)
//   numbers per cell; E, T, P, pi**, bulkPi, muB multiplied by hbar*c.
//   cell_arrays23 order: T P E tau eta ux uy un dat dax day dan pixx pixy pixn piyy piyn bulkPi muB nB Vx Vy Vn
//   (argument order of calculate_dN_pTdpTdphidy after the species arrays, emissionfunction.h:179).
// ---------------------------------------------------------------------------------------------
// (text: the file's bytes; XY: optional {x, y} arrays for the cell positions, columns 2 and 3 -- the sampler wants them)
static int surface_read_vh_text(const std::string &text, const char *path, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                                int32_t dimension, int64_t *n_cells, double *const *A, double *avg5, double *const *XY)
{
    int64_t rows;
    int32_t cols;
    std::vector<double> tab;
    bool exact = false;
    int rc = parse_table(text, path, &rows, &cols, A ? &tab : nullptr, &exact);
    if (rc) return rc;
    if (!A) { *n_cells = rows; return IS3D_OK; }
    if (*n_cells < rows) return io_fail(IS3D_EINVAL, "%s: arrays hold %lld cells, file has %lld", path, (long long)*n_cells, (long long)rows);
    *n_cells = rows;
    enum { iT, iP, iE, itau, ieta, iux, iuy, iun, idat, idax, iday, idan, ipixx, ipixy, ipixn, ipiyy, ipiyn, ibulk, imuB, inB, iVx, iVy, iVn };
    for (int a = 0; a <= ibulk; a++)
        if (!A[a]) return io_fail(IS3D_EINVAL, "cell array %d is NULL", a);
    if (include_baryon && !A[imuB]) return io_fail(IS3D_EINVAL, "include_baryon needs the muB array");
    if (include_baryondiff_deltaf && (!A[inB] || !A[iVx] || !A[iVy] || !A[iVn])) return io_fail(IS3D_EINVAL, "baryon diffusion needs nB, Vx, Vy, Vn arrays");

    const char *p = text.c_str();
    bool short_read = false;
    size_t ti = 0;
    auto next = [&]() -> double {   // the whitespace token stream (surfdat >> ...): the parsed table when it IS that stream
        if (exact) {
            if (ti >= tab.size()) { short_read = true; return 0.0; }
            return tab[ti++];
        }
        char *q;
        double x = strtod(p, &q);
        if (q == p) { short_read = true; return 0.0; }
        p = q;
        return x;
    };
    double Tavg = 0, Eavg = 0, Pavg = 0, muBavg = 0, nBavg = 0, vol = 0;
    for (int64_t i = 0; i < rows; i++) {
        double tau = next();
        const double xpos = next(), ypos = next();
        if (XY) { XY[0][i] = xpos; XY[1][i] = ypos; }
        double eta = next();
        double dat = next(), dax = next(), day = next(), dan = next();
        double ux = next(), uy = next(), un = next();
        double E = next() * kHbarC, T = next() * kHbarC, P = next() * kHbarC;
        double pixx = next() * kHbarC, pixy = next() * kHbarC, pixn = next() * kHbarC;
        double piyy = next() * kHbarC, piyn = next() * kHbarC, bulkPi = next() * kHbarC;
        double muB = 0.0, nB = 0.0;
        if (include_baryon) { muB = next() * kHbarC; A[imuB][i] = muB; }
        if (include_baryondiff_deltaf) {
            nB = next();
            A[inB][i] = nB;
            A[iVx][i] = next();
            A[iVy][i] = next();
            A[iVn][i] = next();
        }
        if (short_read) return io_fail(IS3D_EIO, "%s: ran out of numbers at cell %lld", path, (long long)i);
        (void)dimension;  // the reference only prints a warning for dan != 0 in 2+1D (readindata.cpp:355-359)
        A[itau][i] = tau; A[ieta][i] = eta;
        A[idat][i] = dat; A[idax][i] = dax; A[iday][i] = day; A[idan][i] = dan;
        A[iux][i] = ux; A[iuy][i] = uy; A[iun][i] = un;
        A[iE][i] = E; A[iT][i] = T; A[iP][i] = P;
        A[ipixx][i] = pixx; A[ipixy][i] = pixy; A[ipixn][i] = pixn; A[ipiyy][i] = piyy; A[ipiyn][i] = piyn;
        A[ibulk][i] = bulkPi;
        // surface-volume weighted averages, readindata.cpp:422-450
        double ut = sqrt(1.0 + ux * ux + uy * uy + tau * tau * un * un);
        double udsigma = ut * dat + ux * dax + uy * day + un * dan;
        double dsigma_dsigma = dat * dat - dax * dax - day * day - dan * dan / (tau * tau);
        double mag = fabs(udsigma) + sqrt(fabs(udsigma * udsigma - dsigma_dsigma));
        vol += mag;
        Eavg += E * mag; Tavg += T * mag; Pavg += P * mag; muBavg += muB * mag; nBavg += nB * mag;
    }
    if (avg5) {
        avg5[0] = Tavg / vol; avg5[1] = Eavg / vol; avg5[2] = Pavg / vol; avg5[3] = muBavg / vol; avg5[4] = nBavg / vol;
    }
    return IS3D_OK;
}

extern "C" int is3d_surface_read_vh(const char *path, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                                    int32_t dimension, int64_t *n_cells, double *const *A, double *avg5)
{
    if (!path || !n_cells) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    return surface_read_vh_text(text, path, include_baryon, include_baryondiff_deltaf, dimension, n_cells, A, avg5, nullptr);
}

// ---------------------------------------------------------------------------------------------
// FO_data_reader::read_surf_switch (src/cpp/readindata.cpp:133-144) for the viscous-hydro formats the smooth
// path accepts (emissionfunction.cpp:1503: MODE 0, 1, 4, 6, 7; 5 = vorticity/polarisation and 2, 3 = VAH are
// other paths).  Token stream per cell, converted to the kernel's conventions exactly as the reference does:
//   0  read_surf_VH_old         :148-318   26 cols: tau x y eta | dat dax day dan | ut ux uy un | E T P |
//                                          pitt pitx pity pitn pixx pixy pixn piyy piyn pinn | bulkPi [muB][nB Vt Vx Vy Vn]
//   1  read_surf_VH             :320-468   20 cols (see is3d_surface_read_vh)
//   4  read_surf_VH_MUSIC       :552-668   27 cols: tau x y eta(->0) | d0..d3 (x tau; d3 -> 0 in 2+1D) | ut ux uy un(/tau) |
//                                          E T muB s (P = s T - E) | 10 pi (x hbarc; n components / tau, nn / tau^2) | bulkPi
//   6  read_surf_VH_MUSIC_New   :671-810   29 cols: as 4 with muS muC after muB, eta and d3 ignored (-> 0)
//   7  read_surf_VH_hiceventgen :1059-1196 26 cols, GeV units: tau x y eta(->0) | da_tau da_x da_y (x tau) da_eta(->0) |
//                                          vx vy vn(->0): u = gamma v | 4 pi^t* ignored, pixx pixy pixz(/tau) piyy piyz(/tau) pizz ignored |
//                                          bulkPi | T E P muB
// hbar*c multiplies E, T, P, pi**, bulkPi, muB in modes 0, 1, 4, 6.  u^tau and the pi^{tau mu}, pi^{eta eta} columns are
// read and dropped: the kernel reconstructs them (smooth_kernels.cpp:133, :166-170).
// ---------------------------------------------------------------------------------------------
static int surface_read_text(const std::string &text, const char *path, int32_t mode, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                             int32_t dimension, int64_t *n_cells, double *const *A, double *avg5, double *const *XY)
{
    if (mode == 1) return surface_read_vh_text(text, path, include_baryon, include_baryondiff_deltaf, dimension, n_cells, A, avg5, XY);
    if (mode != 0 && mode != 4 && mode != 5 && mode != 6 && mode != 7)
        return io_fail(IS3D_EINVAL, "surface mode %d is not a viscous-hydro format of the smooth path (0, 1, 4, 5, 6, 7)", mode);
    int64_t rows;
    int32_t cols;
    std::vector<double> tab;
    bool exact = false;
    int rc = parse_table(text, path, &rows, &cols, A ? &tab : nullptr, &exact);
    if (rc) return rc;
    if (!A) { *n_cells = rows; return IS3D_OK; }
    if (*n_cells < rows) return io_fail(IS3D_EINVAL, "%s: arrays hold %lld cells, file has %lld", path, (long long)*n_cells, (long long)rows);
    *n_cells = rows;
    enum { iT, iP, iE, itau, ieta, iux, iuy, iun, idat, idax, iday, idan, ipixx, ipixy, ipixn, ipiyy, ipiyn, ibulk, imuB, inB, iVx, iVy, iVn };
    for (int a = 0; a <= ibulk; a++)
        if (!A[a]) return io_fail(IS3D_EINVAL, "cell array %d is NULL", a);
    const char *p = text.c_str();
    bool short_read = false;
    size_t ti = 0;
    auto next = [&]() -> double {   // the whitespace token stream (surfdat >> ...): the parsed table when it IS that stream
        if (exact) {
            if (ti >= tab.size()) { short_read = true; return 0.0; }
            return tab[ti++];
        }
        char *q;
        double x = strtod(p, &q);
        if (q == p) { short_read = true; return 0.0; }
        p = q;
        return x;
    };
    double Tavg = 0, Eavg = 0, Pavg = 0, muBavg = 0, nBavg = 0, vol = 0;
    for (int64_t i = 0; i < rows; i++) {
        double tau = next();
        const double xpos = next(), ypos = next();
        if (XY) { XY[0][i] = xpos; XY[1][i] = ypos; }
        double eta = next();
        double dat, dax, day, dan, ux, uy, un, E, T, P, pixx, pixy, pixn, piyy, piyn, bulkPi, muB = 0.0, nB = 0.0;
        double Vx = 0.0, Vy = 0.0, Vn = 0.0;
        if (mode == 0) {
            dat = next(); dax = next(); day = next(); dan = next();
            if (dimension == 2 && dan != 0.0)   // reference: message + exit(-1), readindata.cpp:180-184
                return io_fail(IS3D_EINVAL, "%s: 2+1d boost invariant surface read-in error at cell # %lld: dsigma_eta is not zero", path, (long long)i);
            (void)next();  // ut
            ux = next(); uy = next(); un = next();
            E = next() * kHbarC; T = next() * kHbarC; P = next() * kHbarC;
            for (int k = 0; k < 4; k++) (void)next();  // pitt pitx pity pitn
            pixx = next() * kHbarC; pixy = next() * kHbarC; pixn = next() * kHbarC;
            piyy = next() * kHbarC; piyn = next() * kHbarC;
            (void)next();  // pinn
            bulkPi = next() * kHbarC;
            if (include_baryon) muB = next() * kHbarC;
            if (include_baryondiff_deltaf) {
                nB = next();
                (void)next();  // Vt
                Vx = next(); Vy = next(); Vn = next();
            }
        } else if (mode == 5) {
            // read_surf_VH_Vorticity (readindata.cpp:470-551): the mode-1 columns -- with V^tau inside the baryon-diffusion block, as in
            // mode 0 -- then the six components of the thermal vorticity.  calculate_spectra runs the viscous-hydro kernels on such a
            // surface (emissionfunction.cpp:1503, :1643; its calculate_spin_polzn branch at :1675 is unreachable), which do not read them.
            dat = next(); dax = next(); day = next(); dan = next();
            ux = next(); uy = next(); un = next();
            E = next() * kHbarC; T = next() * kHbarC; P = next() * kHbarC;
            pixx = next() * kHbarC; pixy = next() * kHbarC; pixn = next() * kHbarC;
            piyy = next() * kHbarC; piyn = next() * kHbarC;
            bulkPi = next() * kHbarC;
            if (include_baryon) muB = next() * kHbarC;
            if (include_baryondiff_deltaf) {
                nB = next();
                (void)next();  // Vt
                Vx = next(); Vy = next(); Vn = next();
            }
            for (int k = 0; k < 6; k++) (void)next();   // wtx wty wtn wxy wxn wyn
        } else if (mode == 4 || mode == 6) {
            eta = 0.0;
            dat = next() * tau; dax = next() * tau; day = next() * tau;
            dan = next() * tau;
            if (mode == 6 || dimension == 2) dan = 0.0;   // :586-592 (old format zeroes it in 2+1D), :729-730 (new: always)
            (void)next();  // ut
            ux = next(); uy = next();
            un = next() / tau;
            E = next() * kHbarC; T = next() * kHbarC; muB = next() * kHbarC;
            if (mode == 6) { (void)next(); (void)next(); }   // muS, muC
            P = next() * T - E;                              // entropy density: p = T s - e
            for (int k = 0; k < 4; k++) (void)next();        // pitt pitx pity pitn
            pixx = next() * kHbarC; pixy = next() * kHbarC; pixn = next() * kHbarC / tau;
            piyy = next() * kHbarC; piyn = next() * kHbarC / tau;
            (void)next();  // pinn
            bulkPi = next() * kHbarC;
        } else {  // mode 7, hic-eventgen: already in GeV units
            eta = 0.0;
            dat = next() * tau; dax = next() * tau; day = next() * tau;
            (void)next();
            dan = 0.0;
            double vx = next(), vy = next();
            (void)next();  // vn -> 0
            double ut = sqrt(1.0 / (1.0 - vx * vx - vy * vy));   // :1108-1111
            ux = ut * vx; uy = ut * vy; un = 0.0;
            for (int k = 0; k < 4; k++) (void)next();            // pi^tt pi^tx pi^ty pi^tz
            pixx = next(); pixy = next(); pixn = next() / tau;
            piyy = next(); piyn = next() / tau;
            (void)next();  // pi^zz
            bulkPi = next();
            T = next(); E = next(); P = next(); muB = next();
        }
        if (short_read) return io_fail(IS3D_EIO, "%s: ran out of numbers at cell %lld (mode %d)", path, (long long)i, mode);
        A[itau][i] = tau; A[ieta][i] = eta;
        A[idat][i] = dat; A[idax][i] = dax; A[iday][i] = day; A[idan][i] = dan;
        A[iux][i] = ux; A[iuy][i] = uy; A[iun][i] = un;
        A[iE][i] = E; A[iT][i] = T; A[iP][i] = P;
        A[ipixx][i] = pixx; A[ipixy][i] = pixy; A[ipixn][i] = pixn; A[ipiyy][i] = piyy; A[ipiyn][i] = piyn;
        A[ibulk][i] = bulkPi;
        if (A[imuB]) A[imuB][i] = muB;
        if (A[inB]) A[inB][i] = nB;
        if (A[iVx]) A[iVx][i] = Vx;
        if (A[iVy]) A[iVy][i] = Vy;
        if (A[iVn]) A[iVn][i] = Vn;
        double ut = sqrt(1.0 + ux * ux + uy * uy + tau * tau * un * un);
        double udsigma = ut * dat + ux * dax + uy * day + un * dan;
        double dsigma_dsigma = dat * dat - dax * dax - day * day - dan * dan / (tau * tau);
        double mag = fabs(udsigma) + sqrt(fabs(udsigma * udsigma - dsigma_dsigma));
        vol += mag;
        Eavg += E * mag; Tavg += T * mag; Pavg += P * mag; muBavg += muB * mag; nBavg += nB * mag;
    }
    if (avg5) {
        avg5[0] = Tavg / vol; avg5[1] = Eavg / vol; avg5[2] = Pavg / vol; avg5[3] = muBavg / vol; avg5[4] = nBavg / vol;
    }
    return IS3D_OK;
}

extern "C" int is3d_surface_read(const char *path, int32_t mode, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                                 int32_t dimension, int64_t *n_cells, double *const *A, double *avg5)
{
    if (!path || !n_cells) return io_fail(IS3D_EINVAL, "null argument");
    if (mode != 0 && mode != 1 && mode != 4 && mode != 5 && mode != 6 && mode != 7)
        return io_fail(IS3D_EINVAL, "surface mode %d is not a viscous-hydro format of the smooth path (0, 1, 4, 5, 6, 7)", mode);
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    return surface_read_text(text, path, mode, include_baryon, include_baryondiff_deltaf, dimension, n_cells, A, avg5, nullptr);
}

// ---------------------------------------------------------------------------------------------
// FO_data_reader::read_surf_VAH_PLMatch (mode 2; src/cpp/readindata.cpp:813-928) with aL_fit and R200
// (src/cpp/arsenal.cpp:999-1065): 31 numbers per cell, hbar*c on E, T, P, PL, pi^{mu nu}, W^mu, bulkPi, and the anisotropic
// variables (aL, Lambda) inferred from PL/P by the conformal factorisation fit.
// ---------------------------------------------------------------------------------------------
namespace {
// arsenal.cpp:999-1028: rational fit of alpha_L(PL/Peq), coefficients and operation order as written there
double aL_fit(double x)
{
    const double x2 = x * x, x3 = x2 * x, x4 = x3 * x, x5 = x4 * x, x6 = x5 * x, x7 = x6 * x, x8 = x7 * x, x9 = x8 * x, x10 = x9 * x;
    const double x11 = x10 * x, x12 = x11 * x, x13 = x12 * x, x14 = x13 * x;
    return (2.307660683188896e-22 + 1.7179667824677117e-16 * x + 7.2725449826862375e-12 * x2 + 4.2846163672079405e-8 * x3 + 0.00004757224421671691 * x4 +
            0.011776118846199547 * x5 + 0.7235583305942909 * x6 + 11.582755440134724 * x7 + 44.45243622597357 * x8 + 12.673594148032494 * x9 -
            33.75866652773691 * x10 + 8.04299287188939 * x11 + 1.462901772148128 * x12 - 0.6320131889637761 * x13 + 0.048528166213735346 * x14) /
           (5.595674409987461e-19 + 8.059757191879689e-14 * x + 1.2033043382301483e-9 * x2 + 2.9819348588423508e-6 * x3 + 0.0015212379997299082 * x4 +
            0.18185453852532632 * x5 + 5.466199358534425 * x6 + 40.1581708710626 * x7 + 44.38310108782752 * x8 - 55.213789667214364 * x9 +
            1.5449108423263358 * x10 + 11.636087951096759 * x11 - 4.005934533735304 * x12 + 0.4703844693488544 * x13 - 0.014599143701745957 * x14);
}
// arsenal.cpp:1031-1065: R200(aL) = aL t200(xi), xi = 1/aL^2 - 1; *ok = false where the reference exits ("x is out of bounds!")
double R200(double aL, bool *ok)
{
    const double x = (1.0 / (aL * aL)) - 1.0;
    const double delta = 0.01;
    double t200 = 0.0;
    if (x > delta) t200 = 1.0 + (1.0 + x) * atan(sqrt(x)) / sqrt(x);
    else if (x < -delta && x > -1.0) t200 = 1.0 + (1.0 + x) * atanh(sqrt(-x)) / sqrt(-x);
    else if (x >= -delta && x <= delta)
        t200 = 2.0 + x * (0.6666666666666667 + x * (-0.1333333333333333 + x * (0.05714285714285716 + x * (-0.031746031746031744 + x * (0.020202020202020193 +
               x * (-0.013986013986013984 + (0.010256410256410262 - 0.00784313725490196 * x) * x))))));
    else *ok = false;   // x <= -1 (or NaN)
    return aL * t200;
}
}  // namespace

static int surface_read_vah_text(const std::string &text, const char *path, int32_t dimension, int64_t *n_cells, double *const *A)
{
    int64_t rows;
    int32_t cols;
    std::vector<double> tab;
    bool exact = false;
    int rc = parse_table(text, path, &rows, &cols, A ? &tab : nullptr, &exact);
    if (rc) return rc;
    if (!A) { *n_cells = rows; return IS3D_OK; }
    if (*n_cells < rows) return io_fail(IS3D_EINVAL, "%s: arrays hold %lld cells, file has %lld", path, (long long)*n_cells, (long long)rows);
    *n_cells = rows;
    enum { itau, ieta, iux, iuy, iun, idat, idax, iday, idan, iT, ipitt, ipitx, ipity, ipitn, ipixx, ipixy, ipixn, ipiyy, ipiyn, ipinn, ibulk,
           iWx, iWy, iLambda, iaL, iE, iP, iPL, iWt, iWn, ix, iy };
    for (int a = 0; a <= iaL; a++)
        if (!A[a]) return io_fail(IS3D_EINVAL, "VAH cell array %d is NULL", a);
    const char *p = text.c_str();
    bool short_read = false;
    size_t ti = 0;
    auto next = [&]() -> double {   // the whitespace token stream (surface_data >> ...)
        if (exact) {
            if (ti >= tab.size()) { short_read = true; return 0.0; }
            return tab[ti++];
        }
        char *q;
        double x = strtod(p, &q);
        if (q == p) { short_read = true; return 0.0; }
        p = q;
        return x;
    };
    (void)dimension;   // dan != 0 in 2+1D: the reference prints a warning only (its exit is commented out, :853-857)
    for (int64_t i = 0; i < rows; i++) {
        const double tau = next(), x = next(), y = next(), eta = next();
        const double dat = next(), dax = next(), day = next(), dan = next();
        (void)next();   // ut: read and never used (the kernel recomputes it, smooth_kernels.cpp:2213)
        const double ux = next(), uy = next(), un = next();
        const double E = next(), T = next(), P = next(), PL = next();   // fm^-4, fm^-1
        double pi[10], W[4];
        for (double &v : pi) v = next() * kHbarC;
        for (double &v : W) v = next() * kHbarC;
        const double bulkPi = next() * kHbarC;
        if (short_read) return io_fail(IS3D_EIO, "%s: ran out of numbers at cell %lld (mode 2)", path, (long long)i);
        if (!((PL / P) < 3.0))   // :910-921: "pl is too large, stopping anisotropic variables..." exit(-1)
            return io_fail(IS3D_EINVAL, "%s: cell %lld: PL/P = %.6g is too large for the anisotropic variables (needs < 3)", path, (long long)i, PL / P);
        const double aL = aL_fit(PL / P);
        bool ok = true;
        const double r200 = R200(aL, &ok);
        if (!ok) return io_fail(IS3D_EINVAL, "%s: cell %lld: alpha_L = %.6g is out of bounds for R200", path, (long long)i, aL);
        const double Lambda = T / pow(0.5 * aL * r200, 0.25);
        A[itau][i] = tau; A[ieta][i] = eta; A[iux][i] = ux; A[iuy][i] = uy; A[iun][i] = un;
        A[idat][i] = dat; A[idax][i] = dax; A[iday][i] = day; A[idan][i] = dan;
        A[iT][i] = T * kHbarC;
        for (int k = 0; k < 10; k++) A[ipitt + k][i] = pi[k];
        A[ibulk][i] = bulkPi;
        A[iWx][i] = W[1]; A[iWy][i] = W[2];
        A[iLambda][i] = Lambda * kHbarC; A[iaL][i] = aL;
        if (A[iE]) A[iE][i] = E * kHbarC;
        if (A[iP]) A[iP][i] = P * kHbarC;
        if (A[iPL]) A[iPL][i] = PL * kHbarC;
        if (A[iWt]) A[iWt][i] = W[0];
        if (A[iWn]) A[iWn][i] = W[3];
        if (A[ix]) A[ix][i] = x;
        if (A[iy]) A[iy][i] = y;
    }
    return IS3D_OK;
}

extern "C" int is3d_surface_read_vah(const char *path, int32_t dimension, int64_t *n_cells, double *const *A)
{
    if (!path || !n_cells) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    return surface_read_vah_text(text, path, dimension, n_cells, A);
}

// ---------------------------------------------------------------------------------------------
// is3d_surface: a parsed surface owned by the library -- ONE read and ONE parse of the text (the two-call readers above parse it twice
// and the sampler's position columns a third time), and a binary sidecar `<path>.is3dcache` so that the next run on the same file skips
// the parse altogether (0.9 s of a 2 s run at 1e6 cells; 8 GPUs shrink everything else).  No reference counterpart: the reference parses
// the text with operator>> on every run (readindata.cpp:320-468); what is kept is its result -- the cache holds exactly the arrays and
// the averages the text parse produced, bit for bit.
//   sidecar = 128-byte header + the stored arrays as raw fp64, in array order.  It is used only if ALL of these match the text file as it
//   is now: size, mtime (ns), a hash of sampled blocks of its contents (first / last 64 KiB + 256 evenly spaced 4 KiB blocks; the whole
//   file with cache = 2), and the parse parameters (mode, include_baryon, include_baryondiff_deltaf, dimension) -- and if its own size is
//   what the header says.  Anything else: ignored, the text is parsed, the sidecar rewritten (temp file + rename; a directory that
//   cannot be written to is not an error).  IS3D_NO_CACHE=1 (or cache = 0) neither reads nor writes one.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kSurfArraysVH = 25;    // cell_arrays23 + x, y
constexpr int kSurfArraysVAH = 32;   // arrays32 of is3d_surface_read_vah

struct CacheHeader {
    char magic[8];
    uint32_t version, header_bytes;
    uint64_t text_size;
    int64_t text_mtime_ns;
    uint64_t sample_hash, full_hash;       // full_hash 0: not computed when the cache was written
    int32_t mode, include_baryon, include_diff, dimension;
    int64_t n_cells;
    uint64_t array_mask;
    double avg[5];
    uint64_t stat_key;                     // version 2: mix of the text's st_ctim (ns), st_ino, st_dev -- utime() can restore an mtime, nothing restores a ctime
};
static_assert(sizeof(CacheHeader) == 128, "cache header is 128 bytes");
const char kCacheMagic[8] = {'I', 'S', '3', 'D', 'S', 'R', 'F', '1'};

inline uint64_t mix64(uint64_t h, uint64_t w)
{
    h ^= w;
    h *= 0x9E3779B97F4A7C15ULL;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ULL;
    h ^= h >> 32;
    return h;
}
uint64_t hash_bytes(uint64_t h, const char *p, size_t n)
{
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = mix64(h, w);
    }
    uint64_t w = 0;
    if (i < n) memcpy(&w, p + i, n - i);
    return mix64(h, w ^ ((uint64_t)n << 56));
}
// sampled-content hash of an open file of `size` bytes (pread; no dependence on the file position)
bool sample_hash_fd(int fd, uint64_t size, uint64_t *out)
{
    uint64_t h = mix64(0x1553D5AF0C0FFEEULL, size);
    std::vector<char> buf(65536);
    auto block = [&](uint64_t off, size_t len) {
        if (off >= size) return true;
        len = (size_t)std::min<uint64_t>(len, size - off);
        size_t got = 0;
        while (got < len) {
            const ssize_t r = pread(fd, buf.data() + got, len - got, (off_t)(off + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        h = hash_bytes(mix64(h, off), buf.data(), len);
        return true;
    };
    if (!block(0, 65536)) return false;
    if (size > 65536 && !block(size - std::min<uint64_t>(size, 65536), 65536)) return false;
    if (size > 2 * 65536)
        for (int k = 1; k <= 256; k++)
            if (!block((size / 257) * (uint64_t)k, 4096)) return false;
    *out = h;
    return true;
}

}  // namespace

struct is3d_surface {
    int64_t n = 0;
    int32_t mode = 1, include_baryon = 0, include_diff = 0, dimension = 3;
    std::vector<std::vector<double>> a;   // kSurfArraysVH or kSurfArraysVAH arrays; empty = not present
    double avg[5] = {0, 0, 0, 0, 0};
    int32_t source = 0;                   // 0 text parsed (no cache written) | 1 text parsed, cache written | 2 loaded from the cache
    std::thread writer;                   // the sidecar is written while the caller computes
    std::atomic<int> written{0};
    ~is3d_surface() { if (writer.joinable()) writer.join(); }
};

namespace {

bool cache_write(const std::string &cpath, const CacheHeader &hd, const is3d_surface *s)
{
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp.%ld", cpath.c_str(), (long)getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f) return false;
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1;
    for (size_t k = 0; ok && k < s->a.size(); k++)
        if ((hd.array_mask >> k) & 1ULL) ok = s->a[k].empty() || fwrite(s->a[k].data(), sizeof(double), s->a[k].size(), f) == s->a[k].size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp, cpath.c_str()) == 0;
    if (!ok) (void)remove(tmp);
    return ok;
}

}  // namespace

extern "C" int is3d_surface_open(const char *path, int32_t mode, int32_t include_baryon, int32_t include_baryondiff_deltaf, int32_t dimension,
                                 int32_t cache, is3d_surface **out)
{
    if (!path || !out) return io_fail(IS3D_EINVAL, "null argument");
    *out = nullptr;
    const bool vah = mode == 2;
    if (!vah && mode != 0 && mode != 1 && mode != 4 && mode != 5 && mode != 6 && mode != 7)
        return io_fail(IS3D_EINVAL, "surface mode %d: the smooth path reads the formats 0, 1, 2, 4, 5, 6, 7", mode);
    if (cache < 0 || cache > 2) return io_fail(IS3D_EINVAL, "is3d_surface_open: cache = 0 (off) | 1 (use / write the sidecar) | 2 (as 1, whole-file hash)");
    if (const char *e = getenv("IS3D_NO_CACHE"))
        if (*e && strcmp(e, "0")) cache = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    struct stat stt;
    if (fstat(fd, &stt) != 0) { close(fd); return io_fail(IS3D_EIO, "%s: cannot stat", path); }
    const uint64_t tsize = (uint64_t)stt.st_size;
    const int64_t tmtime = (int64_t)stt.st_mtim.tv_sec * 1000000000LL + (int64_t)stt.st_mtim.tv_nsec;
    // an edit that keeps size and mtime (fixed-width hydro output restored by cp -p / rsync -t / utime) and misses every sampled block would
    // otherwise be served from the sidecar: the inode's change time cannot be set from user space, so it joins the key (a copy of the text with
    // its sidecar is re-parsed once: the safe direction)
    const uint64_t tkey = mix64(mix64(mix64(0x1553D5ULL, (uint64_t)stt.st_ctim.tv_sec * 1000000000ULL + (uint64_t)stt.st_ctim.tv_nsec), (uint64_t)stt.st_ino), (uint64_t)stt.st_dev);
    std::unique_ptr<is3d_surface> S(new is3d_surface);
    S->mode = mode; S->include_baryon = include_baryon != 0; S->include_diff = include_baryondiff_deltaf != 0; S->dimension = dimension;
    const int narr = vah ? kSurfArraysVAH : kSurfArraysVH;
    S->a.resize((size_t)narr);
    uint64_t mask = 0;
    for (int k = 0; k < narr; k++) {
        bool used = true;
        if (!vah) used = k < 18 || (k == 18 && S->include_baryon) || (k > 18 && k < 23 && S->include_diff) || k >= 23;
        if (used) mask |= 1ULL << k;
    }
    const std::string cpath = std::string(path) + ".is3dcache";
    uint64_t shash = 0, fhash = 0;
    std::string text;
    if (cache) {
        if (!sample_hash_fd(fd, tsize, &shash)) { close(fd); return io_fail(IS3D_EIO, "%s: read error", path); }
        if (cache == 2) {
            if (!slurp(path, text)) { close(fd); return io_fail(IS3D_EIO, "the data file %s cannot be opened", path); }
            fhash = hash_bytes(0x1553D, text.data(), text.size());
            if (!fhash) fhash = 1;
        }
        // ---- try the sidecar ----
        FILE *cf = fopen(cpath.c_str(), "rb");
        if (cf) {
            CacheHeader hd;
            bool ok = fread(&hd, sizeof hd, 1, cf) == 1 && !memcmp(hd.magic, kCacheMagic, 8) && hd.version == 2 && hd.header_bytes == sizeof hd &&
                      hd.text_size == tsize && hd.text_mtime_ns == tmtime && hd.stat_key == tkey && hd.sample_hash == shash && hd.mode == mode &&
                      hd.include_baryon == S->include_baryon && hd.include_diff == S->include_diff && hd.dimension == dimension &&
                      hd.array_mask == mask && hd.n_cells >= 0 && (cache != 2 || hd.full_hash == fhash);
            if (ok) {
                // n_cells is bounded by the sidecar's own size BEFORE it is multiplied: a corrupt or foreign header cannot wrap the product
                struct stat cst;
                const uint64_t per_cell = (uint64_t)__builtin_popcountll(mask) * sizeof(double);
                ok = fstat(fileno(cf), &cst) == 0 && (uint64_t)cst.st_size >= sizeof hd && per_cell > 0 &&
                     (uint64_t)hd.n_cells <= ((uint64_t)cst.st_size - sizeof hd) / per_cell &&
                     (uint64_t)cst.st_size == sizeof hd + per_cell * (uint64_t)hd.n_cells;
            }
            if (ok) {
                try {   // "a sidecar that cannot be used is ignored": that includes one whose arrays cannot be allocated
                    for (int k = 0; ok && k < narr; k++)
                        if ((mask >> k) & 1ULL) {
                            S->a[(size_t)k].resize((size_t)hd.n_cells);
                            ok = hd.n_cells == 0 || fread(S->a[(size_t)k].data(), sizeof(double), (size_t)hd.n_cells, cf) == (size_t)hd.n_cells;
                        }
                } catch (const std::exception &) {
                    ok = false;
                }
                if (ok) {
                    S->n = hd.n_cells;
                    memcpy(S->avg, hd.avg, sizeof S->avg);
                    S->source = 2;
                }
            }
            fclose(cf);
            if (ok) {
                close(fd);
                *out = S.release();
                return IS3D_OK;
            }
            for (auto &v : S->a) v.clear();
        }
    }
    close(fd);
    // ---- parse the text: one read, one parse ----
    if (text.empty() && !slurp(path, text)) return io_fail(IS3D_EIO, "the data file %s cannot be opened", path);
    int64_t rows = 0;
    for (const char *p = text.data(), *end = p + text.size(); p < end;) {   // a row is a '\n'-terminated line (Table / readBlockData)
        const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!q) break;
        rows++;
        p = q + 1;
    }
    std::vector<double *> ptr((size_t)narr, nullptr);
    try {
        for (int k = 0; k < narr; k++)
            if ((mask >> k) & 1ULL) { S->a[(size_t)k].assign((size_t)std::max<int64_t>(rows, 1), 0.0); ptr[(size_t)k] = S->a[(size_t)k].data(); }
    } catch (const std::exception &) {   // nothing may unwind through the C ABI
        return io_fail(IS3D_ENOMEM, "%s: %lld rows x %d arrays do not fit the host memory", path, (long long)rows, __builtin_popcountll(mask));
    }
    int64_t n = rows;
    int rc;
    if (vah) rc = surface_read_vah_text(text, path, dimension, &n, ptr.data());
    else rc = surface_read_text(text, path, mode, include_baryon, include_baryondiff_deltaf, dimension, &n, ptr.data(), S->avg, ptr.data() + 23);
    if (rc) return rc;
    S->n = n;
    if (n == 0)
        for (auto &v : S->a) v.clear();
    { std::string().swap(text); }
    if (cache) {
        CacheHeader hd;
        memset(&hd, 0, sizeof hd);
        memcpy(hd.magic, kCacheMagic, 8);
        hd.version = 2; hd.header_bytes = sizeof hd;
        hd.text_size = tsize; hd.text_mtime_ns = tmtime; hd.stat_key = tkey; hd.sample_hash = shash; hd.full_hash = fhash;
        hd.mode = mode; hd.include_baryon = S->include_baryon; hd.include_diff = S->include_diff; hd.dimension = dimension;
        hd.n_cells = n; hd.array_mask = mask;
        memcpy(hd.avg, S->avg, sizeof hd.avg);
        is3d_surface *raw = S.get();
        S->source = 1;   // provisional; is3d_surface_source joins the writer and reports what happened
        S->writer = std::thread([raw, cpath, hd] { raw->written.store(cache_write(cpath, hd, raw) ? 1 : -1); });
    }
    *out = S.release();
    return IS3D_OK;
}

extern "C" int64_t is3d_surface_cells(const is3d_surface *s) { return s ? s->n : -1; }

extern "C" int32_t is3d_surface_source(is3d_surface *s)
{
    if (!s) return -1;
    if (s->writer.joinable()) s->writer.join();
    if (s->source == 1 && s->written.load() != 1) s->source = 0;   // the sidecar could not be written (read-only directory): not an error
    return s->source;
}

extern "C" int32_t is3d_surface_from_sidecar(const is3d_surface *s) { return (s && s->source == 2) ? 1 : 0; }

extern "C" int is3d_surface_arrays(const is3d_surface *s, const double **arrays, int32_t n_arrays, double avg5[5])
{
    if (!s || !arrays) return io_fail(IS3D_EINVAL, "null argument");
    if (n_arrays != (int32_t)s->a.size())
        return io_fail(IS3D_EINVAL, "is3d_surface_arrays: this surface (mode %d) has %zu arrays, %d asked for", s->mode, s->a.size(), n_arrays);
    for (size_t k = 0; k < s->a.size(); k++) arrays[k] = s->a[k].empty() ? nullptr : s->a[k].data();
    if (avg5) memcpy(avg5, s->avg, sizeof s->avg);
    return IS3D_OK;
}

extern "C" void is3d_surface_close(is3d_surface *s) { delete s; }

// ---------------------------------------------------------------------------------------------
// The anisotropic-hydro branch of load_df_coefficient_data in the CUDA tree (src/cuda/deltafReader.cu): file names :74-81,
// fscanf("%d\n%d\n") of the two dimensions :104-112, fgets(header, 100) :122-127, then for i2 (alpha_L) outer, i1 (Lambda) inner
// fscanf("%lf\t\t%lf\t\t%lf\n", &L_array[i1], &aL_array[i2], &c[i1][i2]) on the five files in turn (:196-213): every row
// overwrites its node entries, the c4 file's scan comes last.
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_vah_df_read(const char *dir, int32_t *n_L, int32_t *n_aL, double *L, double *aL, double *c, int64_t capacity)
{
    if (!dir || !n_L || !n_aL) return io_fail(IS3D_EINVAL, "null argument");
    long nL0 = -1, naL0 = -1;
    for (int k = 0; k < 5; k++) {
        const std::string path = std::string(dir) + "/c" + std::to_string(k) + "_vah1.dat";
        std::string text;
        if (!slurp(path.c_str(), text)) return io_fail(IS3D_EIO, "Couldn't open c%d coefficient file %s", k, path.c_str());
        const char *p = text.c_str();
        char *q;
        const long nL = strtol(p, &q, 10);
        if (q == p || nL < 2) return io_fail(IS3D_EIO, "%s: bad Lambda dimension", path.c_str());
        p = q;
        const long naL = strtol(p, &q, 10);
        if (q == p || naL < 2) return io_fail(IS3D_EIO, "%s: bad alpha_L dimension", path.c_str());
        p = q;
        if (k == 0) { nL0 = nL; naL0 = naL; }
        else if (nL != nL0 || naL != naL0)   // the reference reads all five headers into the same two ints: the last one would win silently
            return io_fail(IS3D_EINVAL, "%s: dimensions %ld x %ld differ from c0's %ld x %ld", path.c_str(), nL, naL, nL0, naL0);
        if (!L) continue;
        if ((int64_t)5 * nL * naL > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %lld < %lld", path.c_str(), (long long)capacity, (long long)(5 * nL * naL));
        while (*p && isspace((unsigned char)*p)) p++;                         // fscanf("%d\n%d\n") eats the white space
        for (int got = 0; *p && got < 99; got++) { if (*p++ == '\n') break; }   // fgets(header, 100, file): at most 99 characters
        double *tab = c + (size_t)k * nL * naL;
        for (long i2 = 0; i2 < naL; i2++)
            for (long i1 = 0; i1 < nL; i1++) {
                double v[3];
                for (double &x : v) {
                    x = strtod(p, &q);
                    if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path.c_str(), i2 * nL + i1);
                    p = q;
                }
                L[i1] = v[0];
                if (aL) aL[i2] = v[1];
                tab[i2 * nL + i1] = v[2];
            }
    }
    *n_L = (int32_t)nL0;
    *n_aL = (int32_t)naL0;
    if (L) {
        for (long i = 1; i < nL0; i++)
            if (!(L[i] > L[i - 1])) return io_fail(IS3D_EINVAL, "%s: Lambda nodes do not ascend at %ld", dir, i);
        if (aL)
            for (long i = 1; i < naL0; i++)
                if (!(aL[i] > aL[i - 1])) return io_fail(IS3D_EINVAL, "%s: alpha_L nodes do not ascend at %ld", dir, i);
    }
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// PDG_Data::read_resonances_conventional  (src/cpp/readindata.cpp:1440-1568)
//   token stream: 12 header fields + decays x 8 fields; an antiparticle entry follows each
//   baryon > 0; the entry produced by the read attempt that hits EOF is dropped (Nparticle =
//   count - 1, :1540), which for a file WITHOUT trailing whitespace drops the last real entry;
//   sign = (baryon % 2 == 0) ? -1 : +1  (:1544-1545).
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_pdg_read(const char *path, int32_t *n, int64_t *mc_id, double *mass, double *gspin, double *baryon,
                             double *sign, int32_t capacity)
{
    if (!path || !n) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "cannot open %s", path);
    struct Ent { long id; double m; int g, b; };
    std::vector<Ent> v;
    std::vector<std::string> tok;
    {
        std::istringstream ss(text);
        std::string t;
        while (ss >> t) tok.push_back(t);
    }
    const bool trailing_ws = !text.empty() && isspace((unsigned char)text.back());
    size_t i = 0;
    while (i < tok.size()) {
        if (i + 12 > tok.size()) return io_fail(IS3D_EIO, "%s: truncated particle record", path);
        Ent e;
        e.id = strtol(tok[i].c_str(), nullptr, 10);
        e.m = strtod(tok[i + 2].c_str(), nullptr);
        e.g = (int)strtol(tok[i + 4].c_str(), nullptr, 10);
        e.b = (int)strtol(tok[i + 5].c_str(), nullptr, 10);
        int decays = (int)strtol(tok[i + 11].c_str(), nullptr, 10);
        if (decays < 0 || decays > 50) return io_fail(IS3D_EIO, "%s: particle %ld has %d decay channels (max 50)", path, e.id, decays);
        i += 12 + 8 * (size_t)decays;
        if (i > tok.size()) return io_fail(IS3D_EIO, "%s: truncated decay table of particle %ld", path, e.id);
        v.push_back(e);
        if (e.b > 0) v.push_back(Ent{-e.id, e.m, e.g, -e.b});
    }
    if (!trailing_ws && !v.empty()) v.pop_back();
    *n = (int32_t)v.size();
    if (!mc_id) return IS3D_OK;
    if ((int32_t)v.size() > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %d < %zu particles", path, capacity, v.size());
    for (size_t k = 0; k < v.size(); k++) {
        mc_id[k] = v[k].id;
        if (mass) mass[k] = v[k].m;
        if (gspin) gspin[k] = v[k].g;
        if (baryon) baryon[k] = v[k].b;
        if (sign) sign[k] = (v[k].b % 2 == 0) ? -1.0 : 1.0;
    }
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// PDG_Data::read_resonances_smash_box  (src/cpp/readindata.cpp:1571-1685; hrg_eos = 3, PDG/pdg_box.dat) with read_mcid (:1201-1418).
//   Line oriented: "name mass width parity id [id [id [id]]]", '#' starts a comment line (and ends a data line: the extraction of the ids
//   stops at the first token that is not a number), blank lines skipped.  Every non-zero id gives an entry, followed by its antiparticle when
//   the particle has a distinct one.  Spin degeneracy, baryon number, statistics and "has an antiparticle" come from the digits of the
//   Monte-Carlo id: n nR nL nq1 nq2 nq3 nJ (an eighth digit adds to nJ); hadron = nq2, nq3 != 0; meson = nq1 == 0; gspin = nJ;
//   antiparticle iff baryon or nq2 != nq3.  The reference handles the deuteron (1000010020) and non-hadrons with an error print;
//   here they are an error (IS3D_EIO): neither occurs in the shipped file.
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_pdg_read_box(const char *path, int32_t *n, int64_t *mc_id, double *mass, double *gspin, double *baryon,
                                 double *sign, int32_t capacity)
{
    if (!path || !n) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "cannot open %s", path);
    struct Ent { long id; double m; int g, b, s; };
    std::vector<Ent> v;
    std::istringstream all(text);
    std::string line;
    int lineno = 0;
    while (std::getline(all, line)) {
        lineno++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string name;
        double m = 0.0, width = 0.0;
        char parity = 0;
        if (!(ls >> name >> m >> width >> parity)) {
            // (a line of white space only: the reference's extraction fails and adds nothing)
            bool blank = true;
            for (char c : line) blank = blank && isspace((unsigned char)c);
            if (blank) continue;
            return io_fail(IS3D_EIO, "%s:%d: expected \"name mass width parity id...\"", path, lineno);
        }
        for (int k = 0; k < 4; k++) {   // mcid_entries = 4 (:1577)
            long id = 0;
            if (!(ls >> id)) break;
            if (id == 0) continue;
            if (id < 0) return io_fail(IS3D_EIO, "%s:%d: antiparticle id %ld (the file lists particles only)", path, lineno, id);
            int d[10];
            long x = id;
            for (int i = 0; i < 10; i++) { d[i] = (int)(x % 10); x /= 10; }
            if (x > 0) return io_fail(IS3D_EIO, "%s:%d: id %ld has more than 10 digits", path, lineno, id);
            const int nJ = d[0] + d[7], nq3 = d[1], nq2 = d[2], nq1 = d[3];
            const bool hadron = id != 1000010020L && nq3 != 0 && nq2 != 0;
            if (!hadron || nJ <= 0) return io_fail(IS3D_EIO, "%s:%d: id %ld is not a hadron with a spin digit (the reference prints an error there)", path, lineno, id);
            const bool is_baryon = nq1 != 0;
            const Ent e{id, m, nJ, is_baryon ? 1 : 0, is_baryon ? 1 : -1};
            v.push_back(e);
            if (is_baryon || nq2 != nq3) v.push_back(Ent{-id, m, nJ, -e.b, e.s});
        }
    }
    *n = (int32_t)v.size();
    if (!mc_id) return IS3D_OK;
    if ((int32_t)v.size() > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %d < %zu particles", path, capacity, v.size());
    for (size_t k = 0; k < v.size(); k++) {
        mc_id[k] = v[k].id;
        if (mass) mass[k] = v[k].m;
        if (gspin) gspin[k] = v[k].g;
        if (baryon) baryon[k] = v[k].b;
        if (sign) sign[k] = v[k].s;
    }
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Deltaf_Data::load_df_coefficient_data, one file, include_baryon = 0  (src/cpp/deltafReader.cpp:120-197)
//   line 1: points_T, line 2: points_muB, line 3: labels, then rows "T muB value" with T fastest;
//   only the first points_T rows (muB = 0) are kept.
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_df_table_read(const char *path, int32_t *n_T, double *T, double *value, int32_t capacity)
{
    if (!path || !n_T) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "couldn't open coefficient file %s", path);
    const char *p = text.c_str();
    char *q;
    long nT = strtol(p, &q, 10);
    if (q == p || nT < 1) return io_fail(IS3D_EIO, "%s: bad T dimension", path);
    p = q;
    long nB = strtol(p, &q, 10);
    if (q == p || nB < 1) return io_fail(IS3D_EIO, "%s: bad muB dimension", path);
    p = q;
    while (*p && isspace((unsigned char)*p)) p++;  // fscanf("%d\n%d\n") eats the white space
    while (*p && *p != '\n') p++;                  // fgets: label line
    *n_T = (int32_t)nT;
    if (!T) return IS3D_OK;
    if (nT > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %d < %ld", path, capacity, nT);
    for (long i = 0; i < nT; i++) {
        double t = strtod(p, &q);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
        (void)strtod(p, &q);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
        double v = strtod(p, &q);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
        T[i] = t;
        if (value) value[i] = v;
    }
    return IS3D_OK;
}

// The same file with all mu_B rows (include_baryon = 1): value[iB * n_T + iT], storage order of
// load_df_coefficient_data (deltafReader.cpp:168-196).
extern "C" int is3d_df_table_read_full(const char *path, int32_t *n_T, int32_t *n_muB, double *T, double *muB, double *value,
                                       int64_t capacity)
{
    if (!path || !n_T || !n_muB) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "couldn't open coefficient file %s", path);
    const char *p = text.c_str();
    char *q;
    long nT = strtol(p, &q, 10);
    if (q == p || nT < 1) return io_fail(IS3D_EIO, "%s: bad T dimension", path);
    p = q;
    long nB = strtol(p, &q, 10);
    if (q == p || nB < 1) return io_fail(IS3D_EIO, "%s: bad muB dimension", path);
    p = q;
    while (*p && isspace((unsigned char)*p)) p++;
    while (*p && *p != '\n') p++;
    *n_T = (int32_t)nT;
    *n_muB = (int32_t)nB;
    if (!T) return IS3D_OK;
    if (!muB || !value || (int64_t)nT * nB > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %lld < %ld", path, (long long)capacity, nT * nB);
    for (long ib = 0; ib < nB; ib++)
        for (long i = 0; i < nT; i++) {
            double v3[3];
            for (int c = 0; c < 3; c++) {
                v3[c] = strtod(p, &q);
                if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, ib * nT + i);
                p = q;
            }
            T[i] = v3[0];
            muB[ib] = v3[1];
            value[ib * nT + i] = v3[2];
        }
    return IS3D_OK;
}

// Gauss_Laguerre::load_roots_and_weights (src/cpp/readindata.cpp:24-53): "n_alpha n_points" then n_alpha * n_points
// rows "alpha root weight" (the alpha column is a dummy there too).
extern "C" int is3d_gla_read(const char *path, int32_t *n_alpha, int32_t *n_points, double *root, double *weight, int64_t capacity)
{
    if (!path || !n_alpha || !n_points) return io_fail(IS3D_EINVAL, "null argument");
    std::string text;
    if (!slurp(path, text)) return io_fail(IS3D_EIO, "couldn't open gauss laguerre file %s", path);
    const char *p = text.c_str();
    char *q;
    long na = strtol(p, &q, 10);
    if (q == p || na < 1) return io_fail(IS3D_EIO, "%s: bad alpha count", path);
    p = q;
    long np = strtol(p, &q, 10);
    if (q == p || np < 1) return io_fail(IS3D_EIO, "%s: bad point count", path);
    p = q;
    *n_alpha = (int32_t)na;
    *n_points = (int32_t)np;
    if (!root) return IS3D_OK;
    if (!weight || (int64_t)na * np > capacity) return io_fail(IS3D_EINVAL, "%s: capacity %lld < %ld", path, (long long)capacity, na * np);
    for (long i = 0; i < na * np; i++) {
        (void)strtol(p, &q, 10);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
        root[i] = strtod(p, &q);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
        weight[i] = strtod(p, &q);
        if (q == p) return io_fail(IS3D_EIO, "%s: ran out of numbers at row %ld", path, i);
        p = q;
    }
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// write_dN_pTdpTdphidy_toFile (src/cpp/emissionfunction.cpp:381-450), write_continuous_vn_toFile
// (:1053-1136), write_dN_dy_toFile (:729-772); same order as calculate_spectra calls them (:1678-1686).
// All files are opened in append mode, as in the reference.
// ---------------------------------------------------------------------------------------------
extern "C" int is3d_write_results(const char *dir, int32_t dimension, int32_t npart, const int64_t *mc_id, int32_t npT,
                                  const double *pT, const double *pT_w, int32_t nphi, const double *phi, const double *phi_w,
                                  int32_t ny, const double *y, const double *dN)
{
    if (!dir || !mc_id || !pT || !phi || !dN) return io_fail(IS3D_EINVAL, "null argument");
    if (dimension == 3 && !y) return io_fail(IS3D_EINVAL, "3+1D output needs the y grid");
    const int y_pts = (dimension == 2) ? 1 : ny;
    auto idx = [&](int ipart, int ipT, int iphip, int iy) {
        return (long long)ipart + (long long)npart * ((long long)ipT + (long long)npT * ((long long)iphip + (long long)nphi * (long long)iy));
    };
    auto yval = [&](int iy) { return (dimension == 2) ? 0.0 : y[iy]; };
    const std::string base(dir);
    // The text of a species' block goes to two files; producing it is the whole cost of this function -- 0.6 GB for the 305-species list.
    // Three of a row's four numbers (y, phip, pT) are the same for every species: their text "y\tphip\tpT\t" is formatted ONCE per (y, phi, pT)
    // with the reference's own iostream manipulators (scientific, setprecision(8); setw(5) never pads a 14-character number), and a row is
    // that prefix + the value by std::to_chars(scientific, 8) -- the shortest-path Ryu printf, digit for digit what operator<< prints for a
    // finite double ("inf" / "nan" as well: the value carries no setw) -- + '\n'.  Blocks are formatted by up to 16 threads, 64 species at a
    // time; a worker also writes its species' own file, while the concatenated file takes the finished group on a thread of its own.
    std::vector<std::string> prefix((size_t)y_pts * nphi * npT);
    for (int iy = 0; iy < y_pts; iy++)
        for (int iphip = 0; iphip < nphi; iphip++)
            for (int ipT = 0; ipT < npT; ipT++) {
                std::ostringstream f;
                f << std::scientific << std::setw(5) << std::setprecision(8) << yval(iy) << "\t" << phi[iphip] << "\t" << pT[ipT] << "\t";
                prefix[((size_t)iy * nphi + iphip) * npT + ipT] = f.str();
            }
    size_t block_bytes = 0;
    for (const std::string &q : prefix) block_bytes += q.size() + 24;
    block_bytes += (size_t)y_pts * nphi;
    auto block = [&](std::string &out, int ipart) {
        out.clear();
        out.reserve(block_bytes);
        char num[40];
        for (int iy = 0; iy < y_pts; iy++)
            for (int iphip = 0; iphip < nphi; iphip++) {
                for (int ipT = 0; ipT < npT; ipT++) {
                    out += prefix[((size_t)iy * nphi + iphip) * npT + ipT];
                    const auto r = std::to_chars(num, num + sizeof num, dN[idx(ipart, ipT, iphip, iy)], std::chars_format::scientific, 8);
                    out.append(num, (size_t)(r.ptr - num));
                    out.push_back('\n');
                }
                out.push_back('\n');
            }
    };
    {
        // the concatenated file: appended to (the reference opens it in append mode) -- the blocks of a finished group go to their final offsets by
        // pwrite from several threads at once (a single writer copying 0.6 GB into the page cache was the critical path of the whole function)
        const std::string all_path = base + "/dN_pTdpTdphidy.dat";
        const int all_fd = open(all_path.c_str(), O_WRONLY | O_CREAT, 0644);
        if (all_fd < 0) return io_fail(IS3D_EIO, "cannot open %s/dN_pTdpTdphidy.dat (the results directory must exist)", dir);
        struct stat ast;
        int64_t all_off = (fstat(all_fd, &ast) == 0) ? (int64_t)ast.st_size : 0;
        const int group = 64;
        std::vector<std::string> text[2];
        std::vector<std::thread> all_writers;
        std::atomic<int> failed{0};          // 1: a species file could not be opened / written, 2: the concatenated file
        std::atomic<int> failed_part{-1};
        auto join_all_writers = [&] { for (auto &t : all_writers) t.join(); all_writers.clear(); };
        for (int g0 = 0, gi = 0; g0 < npart; g0 += group, gi++) {
            const int g1 = std::min(npart, g0 + group);
            std::vector<std::string> &txt = text[gi & 1];
            txt.resize((size_t)(g1 - g0));
            const int nthreads = (int)std::min<unsigned>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), (unsigned)(g1 - g0));
            auto work = [&](int t) {
                for (int ipart = g0 + t; ipart < g1; ipart += nthreads) {
                    std::string &blk = txt[(size_t)(ipart - g0)];
                    block(blk, ipart);
                    char name[64];
                    snprintf(name, sizeof name, "/dN_pTdpTdphidy_%d.dat", (int)mc_id[ipart]);
                    FILE *f = fopen((base + name).c_str(), "ab");
                    static const char hdr[] = "y\tphip\tpT\tdN_pTdpTdphidy\n";
                    bool ok = f != nullptr;
                    if (ok) ok = fwrite(hdr, 1, sizeof hdr - 1, f) == sizeof hdr - 1 && fwrite(blk.data(), 1, blk.size(), f) == blk.size();
                    if (f && fclose(f) != 0) ok = false;
                    if (!ok) { failed.store(1); failed_part.store(ipart); }
                }
            };
            if (nthreads == 1) work(0);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
                for (auto &x : th) x.join();
            }
            join_all_writers();                    // the previous group is out: its buffer is free for the next group
            std::vector<int64_t> offs(txt.size());
            for (size_t i = 0; i < txt.size(); i++) { offs[i] = all_off; all_off += (int64_t)txt[i].size(); }
            const int nw = (int)std::min<size_t>(8, txt.size());
            for (int w = 0; w < nw; w++)
                all_writers.emplace_back([&txt, offs, w, nw, all_fd, &failed] {
                    for (size_t i = (size_t)w; i < txt.size(); i += (size_t)nw) {
                        const char *p = txt[i].data();
                        size_t left = txt[i].size();
                        int64_t at = offs[i];
                        while (left > 0) {
                            const ssize_t k = pwrite(all_fd, p, left, (off_t)at);
                            if (k <= 0) { failed.store(2); break; }
                            p += k; left -= (size_t)k; at += k;
                        }
                    }
                });
        }
        join_all_writers();
        if (close(all_fd) != 0) failed.store(2);
        if (failed.load() == 1) return io_fail(IS3D_EIO, "cannot write %s/dN_pTdpTdphidy_%d.dat", dir, (int)mc_id[std::max(failed_part.load(), 0)]);
        if (failed.load() == 2) return io_fail(IS3D_EIO, "cannot write %s/dN_pTdpTdphidy.dat", dir);
    }
    if (phi_w) {  // vn_continuous
        const std::complex<double> I(0.0, 1.0);
        const int k_max = 7;
        std::vector<double> ck((size_t)k_max * nphi), sk((size_t)k_max * nphi);   // the same products, evaluated once
        for (int iphip = 0; iphip < nphi; iphip++)
            for (int k = 0; k < k_max; k++) {
                ck[(size_t)iphip * k_max + k] = cos(((double)k + 1.0) * phi[iphip]);
                sk[(size_t)iphip * k_max + k] = sin(((double)k + 1.0) * phi[iphip]);
            }
        for (int ipart = 0; ipart < npart; ipart++) {
            char name[64];
            snprintf(name, sizeof name, "/vn_continuous/vn_%d.dat", (int)mc_id[ipart]);
            std::ofstream f(base + name, std::ios_base::app);
            if (!f) return io_fail(IS3D_EIO, "cannot open %s%s (results/vn_continuous must exist)", dir, name);
            for (int iy = 0; iy < y_pts; iy++) {
                for (int ipT = 0; ipT < npT; ipT++) {
                    double re[k_max] = {0}, im[k_max] = {0}, den = 0.0;
                    for (int iphip = 0; iphip < nphi; iphip++) {
                        double v = dN[idx(ipart, ipT, iphip, iy)];
                        for (int k = 0; k < k_max; k++) {
                            re[k] += ck[(size_t)iphip * k_max + k] * phi_w[iphip] * v;
                            im[k] += sk[(size_t)iphip * k_max + k] * phi_w[iphip] * v;
                        }
                        den += phi_w[iphip] * v;
                    }
                    f << std::scientific << std::setw(5) << std::setprecision(8) << yval(iy) << "\t" << pT[ipT];
                    for (int k = 0; k < k_max; k++) {
                        double vn = std::abs(re[k] + I * im[k]) / den;
                        if (den < 1.e-15) vn = 0.0;
                        f << "\t" << vn;
                    }
                    f << "\n";
                }
                f << "\n";
            }
        }
    }
    if (phi_w && pT_w) {  // dN_dy
        for (int ipart = 0; ipart < npart; ipart++) {
            char name[64];
            snprintf(name, sizeof name, "/dN_dy_%d.dat", (int)mc_id[ipart]);
            std::ofstream f(base + name, std::ios_base::app);
            if (!f) return io_fail(IS3D_EIO, "cannot open %s%s", dir, name);
            for (int iy = 0; iy < y_pts; iy++) {
                double dN_dy = 0.0;
                for (int iphip = 0; iphip < nphi; iphip++)
                    for (int ipT = 0; ipT < npT; ipT++) dN_dy += phi_w[iphip] * pT_w[ipT] * dN[idx(ipart, ipT, iphip, iy)];
                f << std::setw(5) << std::setprecision(8) << yval(iy) << "\t" << dN_dy << "\n";
            }
        }
    }
    return IS3D_OK;
}

// write_particle_list_OSC (src/cpp/emissionfunction.cpp:863-901): one "# N" header per non-empty event (empty events are
// not written: the urqmd afterburner crashes on them), then rows "mcid t x y z E px py pz", scientific, setprecision(16).
extern "C" int is3d_write_particle_list_osc(const char *path, int32_t n_events, int64_t n_particles, const is3d_particle *particles,
                                            const int64_t *mc_id)
{
    if (!path || (n_particles > 0 && (!particles || !mc_id))) return io_fail(IS3D_EINVAL, "null argument");
    std::ofstream f(path, std::ios_base::out);
    if (!f) return io_fail(IS3D_EIO, "couldn't open %s for writing", path);
    int64_t i = 0;
    for (int32_t ev = 0; ev < n_events; ev++) {
        int64_t j = i;
        while (j < n_particles && particles[j].event == ev) j++;
        if (j > i) {
            f << "# " << (j - i) << "\n";
            for (; i < j; i++) {
                const is3d_particle &q = particles[i];
                f << mc_id[q.species] << " " << std::scientific << std::setw(5) << std::setprecision(16) << q.t << " " << q.x << " " << q.y << " "
                  << q.z << " " << q.E << " " << q.px << " " << q.py << " " << q.pz << "\n";
            }
        }
    }
    if (i != n_particles) return io_fail(IS3D_EINVAL, "particles are not ordered by event (entry %lld has event %d)", (long long)i, particles[i].event);
    f.close();
    if (!f) return io_fail(IS3D_EIO, "write error on %s", path);
    return IS3D_OK;
}

// test_sampler = 1: the binned self-consistency outputs of the sampler (sample_dN_dy ... sample_dN_dX,
// emissionfunction_sampling_kernels.cpp:31-152, and their writers, emissionfunction.cpp:903-1257), built from the particle
// list on the host: results/dN_dy/dN_dy_<id>_test.dat (+ _average_test), dN_deta/dN_deta_<id>_test.dat,
// momentum_distribution/dN_2pipTdpTdy_<id>_test.dat, vn/vn_<id>_test.dat, spacetime_distribution/dN_taudtaudy_sampled_<id>_test.dat
// and dN_twopirdrdy_sampled_<id>_test.dat, mean_yield.dat, yield_list.dat.  The directories must exist, as for the reference.
extern "C" int is3d_write_sampler_tests(const char *results_dir, const is3d_sampler_test_bins *b, int32_t n_events, int32_t n_species,
                                        const int64_t *mc_id, int64_t n_particles, const is3d_particle *particles, double mean_yield)
{
    if (!results_dir || !b || !mc_id || (n_particles > 0 && !particles)) return io_fail(IS3D_EINVAL, "null argument");
    if (b->y_bins < 1 || b->eta_bins < 1 || b->pT_bins < 1 || b->tau_bins < 1 || b->r_bins < 1 || n_events < 1 || n_species < 1)
        return io_fail(IS3D_EINVAL, "sampler test bins must be positive");
    const int K_MAX = 7;                                                     // emissionfunction.h:132
    const double two_pi = 2.0 * M_PI;
    const double yw = 2.0 * b->y_cut / (double)b->y_bins, ew = 2.0 * b->eta_cut / (double)b->eta_bins;
    const double pw = (b->pT_upper_cut - b->pT_lower_cut) / (double)b->pT_bins;
    const double tw = (b->tau_max - b->tau_min) / (double)b->tau_bins, rw = (b->r_max - b->r_min) / (double)b->r_bins;
    std::vector<double> dy((size_t)n_species * b->y_bins, 0.0), de((size_t)n_species * b->eta_bins, 0.0), dp((size_t)n_species * b->pT_bins, 0.0);
    std::vector<double> vc((size_t)n_species * b->pT_bins, 0.0), vr((size_t)K_MAX * n_species * b->pT_bins, 0.0), vi(vr.size(), 0.0);
    std::vector<double> dt((size_t)n_species * b->tau_bins, 0.0), dr((size_t)n_species * b->r_bins, 0.0);
    std::vector<int64_t> yield((size_t)n_events, 0);
    for (int64_t i = 0; i < n_particles; i++) {
        const is3d_particle &q = particles[i];
        if (q.species < 0 || q.species >= n_species || q.event < 0 || q.event >= n_events) return io_fail(IS3D_EINVAL, "particle %lld: species or event out of range", (long long)i);
        const int ip = q.species;
        yield[q.event] += 1;
        const double yp = 0.5 * std::log((q.E + q.pz) / (q.E - q.pz));
        const int iyp = (int)std::floor((yp + b->y_cut) / yw);                  // sample_dN_dy
        if (iyp >= 0 && iyp < b->y_bins) dy[(size_t)ip * b->y_bins + iyp] += 1.0;
        const int ieta = (int)std::floor((q.eta + b->eta_cut) / ew);            // sample_dN_deta
        if (ieta >= 0 && ieta < b->eta_bins) de[(size_t)ip * b->eta_bins + ieta] += 1.0;
        if (std::fabs(yp) <= b->y_cut) {
            const double pT = std::sqrt(q.px * q.px + q.py * q.py);
            const int ipT = (int)std::floor((pT - b->pT_lower_cut) / pw);       // sample_dN_2pipTdpTdy, sample_vn
            if (ipT >= 0 && ipT < b->pT_bins) {
                dp[(size_t)ip * b->pT_bins + ipT] += 1.0;
                vc[(size_t)ip * b->pT_bins + ipT] += 1.0;
                double phi = std::atan2(q.py, q.px);
                if (phi < 0.0) phi += 2.0 * M_PI;
                for (int k = 0; k < K_MAX; k++) {
                    vr[((size_t)k * n_species + ip) * b->pT_bins + ipT] += std::cos(((double)k + 1.0) * phi);
                    vi[((size_t)k * n_species + ip) * b->pT_bins + ipT] += std::sin(((double)k + 1.0) * phi);
                }
            }
            const double r = std::sqrt(q.x * q.x + q.y * q.y);                   // sample_dN_dX
            const int itau = (int)std::floor((q.tau - b->tau_min) / tw), ir = (int)std::floor((r - b->r_min) / rw);
            if (itau >= 0 && itau < b->tau_bins) dt[(size_t)ip * b->tau_bins + itau] += 1.0;
            if (ir >= 0 && ir < b->r_bins) dr[(size_t)ip * b->r_bins + ir] += 1.0;
        }
    }
    const std::string root(results_dir);
    const double Nev = (double)n_events;
    for (int ip = 0; ip < n_species; ip++) {
        const std::string id = std::to_string((long long)mc_id[ip]);
        std::ofstream f1(root + "/dN_dy/dN_dy_" + id + "_test.dat"), f2(root + "/dN_dy/dN_dy_" + id + "_average_test.dat");
        std::ofstream f3(root + "/dN_deta/dN_deta_" + id + "_test.dat"), f4(root + "/momentum_distribution/dN_2pipTdpTdy_" + id + "_test.dat");
        std::ofstream f5(root + "/vn/vn_" + id + "_test.dat"), f6(root + "/spacetime_distribution/dN_taudtaudy_sampled_" + id + "_test.dat");
        std::ofstream f7(root + "/spacetime_distribution/dN_twopirdrdy_sampled_" + id + "_test.dat");
        if (!f1 || !f2 || !f3 || !f4 || !f5 || !f6 || !f7)
            return io_fail(IS3D_EIO, "couldn't open the sampler test files under %s (dN_dy/, dN_deta/, momentum_distribution/, vn/, spacetime_distribution/ must exist)", results_dir);
        double avg = 0.0;
        for (int i = 0; i < b->y_bins; i++) {                                    // :919-937
            avg += dy[(size_t)ip * b->y_bins + i];
            f1 << std::setprecision(6) << (-b->y_cut + yw * ((double)i + 0.5)) << "\t" << dy[(size_t)ip * b->y_bins + i] / (yw * Nev) << std::endl;
        }
        f2 << std::setprecision(6) << avg / (2.0 * b->y_cut * Nev) << std::endl;
        for (int i = 0; i < b->eta_bins; i++)                                    // :961-972
            f3 << std::setprecision(6) << (-b->eta_cut + ew * ((double)i + 0.5)) << "\t" << de[(size_t)ip * b->eta_bins + i] / (ew * Nev) << std::endl;
        for (int i = 0; i < b->pT_bins; i++) {                                   // :991-1001, :1156-1177
            const double pT_mid = b->pT_lower_cut + pw * ((double)i + 0.5);
            f4 << std::setprecision(6) << std::scientific << pT_mid << "\t" << dp[(size_t)ip * b->pT_bins + i] / (two_pi * 2.0 * b->y_cut * pw * pT_mid * Nev) << "\n";
            f5 << std::setprecision(6) << std::scientific << pT_mid;
            for (int k = 0; k < K_MAX; k++) {
                const size_t j = ((size_t)k * n_species + ip) * b->pT_bins + i;
                double vn_abs = std::sqrt(vr[j] * vr[j] + vi[j] * vi[j]) / vc[(size_t)ip * b->pT_bins + i];
                if (std::isnan(vn_abs) || std::isinf(vn_abs)) vn_abs = 0.0;
                f5 << "\t" << vn_abs;
            }
            f5 << "\n";
        }
        for (int i = 0; i < b->r_bins; i++) {                                    // :1215-1221
            const double r_mid = b->r_min + rw * ((double)i + 0.5);
            f7 << std::setprecision(6) << std::scientific << r_mid << "\t" << dr[(size_t)ip * b->r_bins + i] / (2.0 * M_PI * r_mid * rw * Nev * 2.0 * b->y_cut) << "\n";
        }
        for (int i = 0; i < b->tau_bins; i++) {                                  // :1223-1229
            const double tau_mid = b->tau_min + tw * ((double)i + 0.5);
            f6 << std::setprecision(6) << std::scientific << tau_mid << "\t" << dt[(size_t)ip * b->tau_bins + i] / (tau_mid * tw * Nev * 2.0 * b->y_cut) << "\n";
        }
    }
    std::ofstream fm(root + "/mean_yield.dat"), fl(root + "/yield_list.dat");   // :1244-1257
    if (!fm || !fl) return io_fail(IS3D_EIO, "couldn't open %s/mean_yield.dat", results_dir);
    fm << mean_yield << std::endl;
    fl << "sampled particle yield\n";
    for (int e = 0; e < n_events; e++) fl << yield[e] << std::endl;
    return IS3D_OK;
}
