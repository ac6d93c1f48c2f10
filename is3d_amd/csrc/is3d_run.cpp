// is3d_run.cpp -- IS3D::run_particlization (/root/reference/src/cpp/iS3D.cpp:74-192) behind the C ABI: the driver layer
// shared by the command line tool (is3d_main.cpp) and the embedding class (include/iS3D_amd.hpp).
//
// Runs in a directory laid out like the reference's run directory; it reads the same hard-coded
// CWD-relative files IS3D::run_particlization reads (/root/reference/src/cpp/iS3D.cpp:74-192):
//   iS3D_parameters.dat, input/surface.dat, PDG/pdg-urqmd_v3.3+.dat | PDG/pdg_smash.dat,
//   PDG/chosen_particles.dat, deltaf_coefficients/vh/<eos>/{c0,c2,F,betabulk,betapi}.dat,
//   tables/pT_gauss_legendre_table.dat, tables/phi_gauss_legendre_table.dat,
//   tables/y_trapezoid_table_21pt.dat, tables/eta/eta_trapezoid_table_241pt.dat
// and writes what EmissionFunctionArray::calculate_spectra writes for operation = 1
// (emissionfunction.cpp:1678-1686) into results/ (which must exist, README.md:34) plus
// average_thermodynamic_quantities.dat (readindata.cpp:464-466).
// operation = 2 (particle sampler; df_mode 1-4, fast in {0, 1}, include_baryon = 1 with df_mode 1-3) writes
// results/particle_list_osc.dat (write_particle_list_OSC, emissionfunction.cpp:863-901) and results/dN_dy_*.dat is skipped.
// mode = 2 (anisotropic hydro, P_L matching; BASELINE config 5): operation = 1 with df_mode = 4 -- the combination for which the
// reference allocates the per-cell c0..c4 (emissionfunction.cpp:1397-1418) and the CUDA tree loads the VAH tables
// (src/cuda/deltafReader.cu:74-81) -- reads input/surface.dat with read_surf_VAH_PLMatch and deltaf_coefficients/vah/c{0..4}_vah1.dat,
// runs what the commented-out call site would (emissionfunction.cpp:1650-1654) and writes the same three result files.
// Scope: operation in {1, 2}, mode in {0, 1, 4, 5, 6, 7}, df_mode in {1, 2, 3} with include_baryon in {0, 1}, df_mode 4
// (modified equilibrium; also reads tables/gla_roots_weights_32_points.txt, deta_min, mass_pion0 and the surface
// averages it has just written, as the reference does) with include_baryon = 0.  Anything else is refused
// with a message instead of silently doing something different from the reference.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iomanip>
#include <string>
#include <thread>
#include <vector>

#include <cstdarg>

#include "../../include/is3d_amd.h"
#include "errors.h"

static double now_s()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// print like the reference does before exit(-1), but return a code and keep the text for is3d_last_error()
static int die(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static int die(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "iS3D-amd: %s\n", buf);
    return is3d::set_error(IS3D_EINVAL, "%s", buf);
}
#define DIE(...) return die(__VA_ARGS__)

static int get_param(const char *name, double *v, bool required = true)
{
    int rc = is3d_param_get("iS3D_parameters.dat", name, v);
    if (rc && required) fprintf(stderr, "iS3D-amd: %s\n", is3d_last_error());
    return rc;
}

static int read_table(const char *path, std::vector<double> &col1, std::vector<double> &col2)
{
    int64_t rows;
    int32_t cols;
    if (is3d_table_read(path, &rows, &cols, nullptr, 0)) return 1;
    std::vector<double> d((size_t)rows * cols);
    if (is3d_table_read(path, &rows, &cols, d.data(), (int64_t)d.size())) return 1;
    col1.resize(rows);
    col2.assign(rows, 0.0);
    for (int64_t r = 0; r < rows; r++) {
        col1[r] = d[r * cols];
        if (cols > 1) col2[r] = d[r * cols + 1];
    }
    return 0;
}

// surface-volume weighted averages of T, E, P, muB, nB as the readers accumulate them (readindata.cpp:422-466)
static void surface_averages(const is3d_cells *c, double avg[5])
{
    double num[5] = {0, 0, 0, 0, 0}, den = 0.0;
    for (int64_t i = 0; i < c->n_cells; i++) {
        const double tau = c->tau[i], ux = c->ux[i], uy = c->uy[i], un = c->un[i];
        const double ut = std::sqrt(1.0 + ux * ux + uy * uy + tau * tau * un * un);
        const double dat = c->dat[i], dax = c->dax[i], day = c->day[i], dan = c->dan[i];
        const double uds = ut * dat + ux * dax + uy * day + un * dan;
        const double dsds = dat * dat - dax * dax - day * day - dan * dan / (tau * tau);
        const double w = std::fabs(uds) + std::sqrt(std::fabs(uds * uds - dsds));
        num[0] += c->T[i] * w; num[1] += c->E[i] * w; num[2] += c->P[i] * w;
        if (c->muB) num[3] += c->muB[i] * w;
        if (c->nB) num[4] += c->nB[i] * w;
        den += w;
    }
    for (int k = 0; k < 5; k++) avg[k] = den > 0.0 ? num[k] / den : 0.0;
}

// devices of the run: an explicit list, else the environment (IS3D_DEVICES = "0,2,3" | "all"; IS3D_REDUCE = "ordered" | "rccl"),
// else every visible device
struct RunDevices {
    std::vector<int32_t> list;   // empty: all visible
    int32_t reduce = IS3D_REDUCE_ORDERED;
};

static int parse_run_devices(const int32_t *devices, int32_t n_devices, int32_t reduce, RunDevices &rd)
{
    rd.reduce = reduce;
    if (devices && n_devices > 0) {
        rd.list.assign(devices, devices + n_devices);
        return IS3D_OK;
    }
    if (n_devices > 0) {
        for (int32_t i = 0; i < n_devices; i++) rd.list.push_back(i);
        return IS3D_OK;
    }
    if (const char *e = getenv("IS3D_REDUCE")) {
        if (!strcmp(e, "rccl")) rd.reduce = IS3D_REDUCE_RCCL;
        else if (!strcmp(e, "ordered")) rd.reduce = IS3D_REDUCE_ORDERED;
        else return die("IS3D_REDUCE = %s: ordered | rccl", e);
    }
    if (const char *e = getenv("IS3D_DEVICES")) {
        if (strcmp(e, "all") && *e) {
            const char *p = e;
            while (*p) {
                char *end = nullptr;
                long v = strtol(p, &end, 10);
                if (end == p || v < 0) return die("IS3D_DEVICES = %s: a comma separated list of HIP device ordinals, or all", e);
                rd.list.push_back((int32_t)v);
                p = (*end == ',') ? end + 1 : end;
                if (*end && *end != ',') return die("IS3D_DEVICES = %s: a comma separated list of HIP device ordinals, or all", e);
            }
        }
    }
    return IS3D_OK;
}

static int run_impl(const is3d_cells *mem, const double *mem_x, const double *mem_y, int variant, const RunDevices &rd, is3d_run_result *res)
{
    printf("iS3D-amd: MI355X-native smooth Cooper-Frye spectra (%s)\n", is3d_version());
    double v;
    int operation, mode, hrg_eos, dimension, df_mode, include_baryon, include_bulk, include_shear, include_diff, regulate, outflow;
#define GET(var, name)                           \
    if (get_param(name, &v)) return IS3D_EINVAL; \
    var = (int)v;
    GET(operation, "operation");
    GET(mode, "mode");
    GET(hrg_eos, "hrg_eos");
    GET(dimension, "dimension");
    GET(df_mode, "df_mode");
    GET(include_baryon, "include_baryon");
    GET(include_bulk, "include_bulk_deltaf");
    GET(include_shear, "include_shear_deltaf");
    GET(include_diff, "include_baryondiff_deltaf");
    GET(regulate, "regulate_deltaf");
    GET(outflow, "outflow");
#undef GET
    if (operation != 1 && operation != 2) DIE("operation = %d: only operation = 1 (smooth momentum spectra) and 2 (particle sampler) are on this path", operation);
    {
        double decays = 0.0;   // optional key here; the reference runs do_resonance_decays() after the spectra (emissionfunction.cpp:1689-1698)
        if (get_param("do_resonance_decays", &decays, false) == IS3D_OK && (int)decays)
            DIE("do_resonance_decays = 1: resonance decays are not on this path; set do_resonance_decays = 0");
    }
    const bool vah = !mem && mode == 2;
    if (vah) {
        if (operation != 1) DIE("mode = 2 (anisotropic hydro): only operation = 1; the reference's VAH sampler is an empty stub (emissionfunction_sampling_kernels.cpp:1231-1239)");
        if (df_mode != 4) DIE("mode = 2 (anisotropic hydro) needs df_mode = 4: the per-cell 14-moment coefficients c0..c4 exist for that combination only (emissionfunction.cpp:1410-1418)");
    }
    if (!mem && !vah && mode != 0 && mode != 1 && mode != 4 && mode != 5 && mode != 6 && mode != 7)
        DIE("mode = %d: the smooth path reads the viscous-hydro surface formats 0, 1, 4, 5, 6, 7 and the anisotropic-hydro format 2 (3 = VAH P_L, P_T matching has no kernel in the reference: emissionfunction.cpp:1328-1681)", mode);
    if (df_mode < 1 || df_mode > 4) DIE("df_mode = %d: 1 (14-moment), 2 (Chapman-Enskog), 3 (modified equilibrium, Mike), 4 (Jonah)", df_mode);
    const bool feqmod = !vah && (df_mode == 3 || df_mode == 4);
    if (df_mode == 4 && include_baryon && !vah)   // deltafReader.cpp:470-474
        DIE("Bilinear interpolation error: Jonah df doesn't work for nonzero muB (df_mode = 4 with include_baryon = 1)");
    const char *pdg_path, *df_dir;
    if (hrg_eos == 1) { pdg_path = "PDG/pdg-urqmd_v3.3+.dat"; df_dir = "deltaf_coefficients/vh/urqmd/"; }
    else if (hrg_eos == 2) { pdg_path = "PDG/pdg_smash.dat"; df_dir = "deltaf_coefficients/vh/smash/"; }
    else if (hrg_eos == 3) { pdg_path = "PDG/pdg_box.dat"; df_dir = "deltaf_coefficients/vh/smash_box/"; }   // readindata.h:219, deltafReader.h:29
    else DIE("hrg_eos = %d: please choose hrg_eos = (1,2,3)", hrg_eos);

    double t0 = now_s();
    // the device contexts are created while the surface is being parsed (joined before the first device call)
    std::vector<int> warm_list(rd.list.begin(), rd.list.end());
    std::thread warm([&warm_list] { is3d::warm_devices(warm_list.empty() ? nullptr : warm_list.data(), (int)warm_list.size()); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } warm_joiner{warm};
    // ---- surface (iS3D.cpp:90-98) ----
    int64_t n_cells = 0;
    double *ptr[23] = {nullptr};
    double avg[5] = {0, 0, 0, 0, 0};
    // the text surface: one read + one parse, or the binary sidecar input/surface.dat.is3dcache of an earlier run on the same file
    // (is3d_surface_open; IS3D_NO_CACHE=1 disables); the arrays are the library's until the end of this function
    is3d_surface *surf = nullptr;
    struct SurfCloser { is3d_surface *&s; ~SurfCloser() { if (s) is3d_surface_close(s); } } surf_closer{surf};
    const double *sa[32] = {nullptr};     // modes 0-7 except 2: cell_arrays23 + x, y; mode 2: arrays32
    if (vah) {
        if (is3d_surface_open("input/surface.dat", 2, 0, 0, dimension, 1, &surf)) DIE("%s", is3d_last_error());
        n_cells = is3d_surface_cells(surf);
        if (is3d_surface_arrays(surf, sa, 32, nullptr)) DIE("%s", is3d_last_error());
        if (is3d_surface_from_sidecar(surf))
            printf("surface: %lld cells from the binary sidecar input/surface.dat.is3dcache (IS3D_NO_CACHE=1 to re-parse the text)\n", (long long)n_cells);
    } else if (mem) {
        // iS3D.cpp:100-134: the surface comes from the caller's vectors, already in GeV / fm units (no hbar*c conversion)
        printf("Reading in freezeout surface from memory \n");
        n_cells = mem->n_cells;
        const double *src[23] = {mem->T, mem->P, mem->E, mem->tau, mem->eta, mem->ux, mem->uy, mem->un, mem->dat, mem->dax, mem->day, mem->dan,
                                 mem->pixx, mem->pixy, mem->pixn, mem->piyy, mem->piyn, mem->bulkPi, mem->muB, mem->nB, mem->Vx, mem->Vy, mem->Vn};
        for (int a = 0; a < 23; a++) ptr[a] = const_cast<double *>(src[a]);
        for (int a = 0; a < 18; a++)
            if (!ptr[a] && !(a == 4 && dimension == 2)) DIE("in-memory surface: a required array is NULL");
        if (n_cells > 0) surface_averages(mem, avg);
    } else {
        if (is3d_surface_open("input/surface.dat", mode, include_baryon, include_diff, dimension, 1, &surf)) DIE("%s", is3d_last_error());
        n_cells = is3d_surface_cells(surf);
        if (is3d_surface_arrays(surf, sa, 25, avg)) DIE("%s", is3d_last_error());
        // said BEFORE the spectra are computed: a user who did not expect the cache sees it while the run can still be stopped
        if (is3d_surface_from_sidecar(surf))
            printf("surface: %lld cells from the binary sidecar input/surface.dat.is3dcache (its key -- size, mtime, ctime, inode, content hash -- matches "
                   "input/surface.dat; IS3D_NO_CACHE=1 to re-parse the text)\n", (long long)n_cells);
        for (int a = 0; a < 23; a++) ptr[a] = const_cast<double *>(sa[a]);   // read-only from here on
    }
    if (!vah) {   // read_surf_VAH_PLMatch accumulates no averages (readindata.cpp:813-928)
        std::ofstream f("average_thermodynamic_quantities.dat", std::ios_base::out);
        f << std::setprecision(15) << avg[0] << "\n" << avg[1] << "\n" << avg[2] << "\n" << avg[3] << "\n" << avg[4];
    }
    // ---- species (iS3D.cpp:138-140, 156; emissionfunction.cpp:336-351, 1293-1307) ----
    int32_t npdg = 0;
    const auto pdg_read = (hrg_eos == 3) ? is3d_pdg_read_box : is3d_pdg_read;   // read_resonances: conventional | smash box (readindata.cpp:1687-1713)
    if (pdg_read(pdg_path, &npdg, nullptr, nullptr, nullptr, nullptr, nullptr, 0)) DIE("%s", is3d_last_error());
    std::vector<int64_t> pid(npdg);
    std::vector<double> pmass(npdg), pg(npdg), pb(npdg), ps(npdg);
    if (pdg_read(pdg_path, &npdg, pid.data(), pmass.data(), pg.data(), pb.data(), ps.data(), npdg)) DIE("%s", is3d_last_error());
    std::vector<double> chosen, dummy;
    if (read_table("PDG/chosen_particles.dat", chosen, dummy)) DIE("%s", is3d_last_error());
    std::vector<int64_t> mcid;
    std::vector<double> mass, sign, deg, bar;
    for (double c : chosen) {
        int id = (int)c;
        bool found = false;
        for (int n = 0; n < npdg; n++)
            if (pid[n] == id) {
                mcid.push_back(pid[n]); mass.push_back(pmass[n]); sign.push_back(ps[n]); deg.push_back(pg[n]); bar.push_back(pb[n]);
                found = true;
                break;
            }
        if (!found) DIE("chosen particle %d is not in %s", id, pdg_path);
    }
    // ---- grids (iS3D.cpp:161-167) ----
    std::vector<double> pT, pTw, phi, phiw, y, yw, eta, etaw;
    if (read_table("tables/pT_gauss_legendre_table.dat", pT, pTw) || read_table("tables/phi_gauss_legendre_table.dat", phi, phiw) ||
        read_table("tables/y_trapezoid_table_21pt.dat", y, yw))
        DIE("%s", is3d_last_error());
    // the eta table: 241 points for the smooth spectra; the reference opens the 41-point one when it samples (iS3D.cpp:164-167) and never uses it
    // there (nor does the sampler here): a run directory made for either operation is accepted
    if (operation == 2) {
        // neither there: the reference fails on its missing table (readBlockData, arsenal.cpp:406-413) -- a mis-built run directory must not pass silently
        if (read_table("tables/eta/eta_trapezoid_table_41pt.dat", eta, etaw) && read_table("tables/eta/eta_trapezoid_table_241pt.dat", eta, etaw))
            DIE("operation 2 opens tables/eta/eta_trapezoid_table_41pt.dat (or tables/eta/eta_trapezoid_table_241pt.dat): neither can be read (%s)", is3d_last_error());
    } else if (read_table("tables/eta/eta_trapezoid_table_241pt.dat", eta, etaw))
        DIE("%s", is3d_last_error());
    // ---- delta-f coefficient tables (iS3D.cpp:144-145) ----
    //      include_baryon = 0: the mu_B = 0 rows; include_baryon = 1: the full (mu_B, T) grids (deltafReader.cpp:134)
    const char *names[10] = {"c0.dat", "c1.dat", "c2.dat", "c3.dat", "c4.dat", "F.dat", "G.dat", "betabulk.dat", "betaV.dat", "betapi.dat"};
    std::vector<double> Tk, Bk, tab[10];
    for (int t = 0; t < 10 && !vah; t++) {
        std::string p = std::string(df_dir) + names[t];
        int32_t nT = 0, nB = 0;
        const bool spline_table = (t == 0 || t == 2 || t == 5 || t == 7 || t == 9);   // c0 c2 F betabulk betapi
        if (!include_baryon && !spline_table) continue;   // only the bilinear branch looks at c1 c3 c4 G betaV
        if (is3d_df_table_read_full(p.c_str(), &nT, &nB, nullptr, nullptr, nullptr, 0)) DIE("%s", is3d_last_error());
        if (!include_baryon) {
            Tk.resize(nT);
            Bk.assign(1, 0.0);
            tab[t].resize(nT);
            if (is3d_df_table_read(p.c_str(), &nT, Tk.data(), tab[t].data(), nT)) DIE("%s", is3d_last_error());
        } else {
            Tk.resize(nT);
            Bk.resize(nB);
            tab[t].resize((size_t)nT * nB);
            if (is3d_df_table_read_full(p.c_str(), &nT, &nB, Tk.data(), Bk.data(), tab[t].data(), (int64_t)tab[t].size())) DIE("%s", is3d_last_error());
        }
    }
    if (warm.joinable()) warm.join();
    double t1 = now_s();
    printf("Total number of freezeout cells: %lld\nNumber of chosen particles: %zu\n", (long long)n_cells, mcid.size());

    is3d_cells cells{};
    cells.n_cells = n_cells;
    cells.T = ptr[0]; cells.P = ptr[1]; cells.E = ptr[2]; cells.tau = ptr[3]; cells.eta = ptr[4];
    cells.ux = ptr[5]; cells.uy = ptr[6]; cells.un = ptr[7];
    cells.dat = ptr[8]; cells.dax = ptr[9]; cells.day = ptr[10]; cells.dan = ptr[11];
    cells.pixx = ptr[12]; cells.pixy = ptr[13]; cells.pixn = ptr[14]; cells.piyy = ptr[15]; cells.piyn = ptr[16];
    cells.bulkPi = ptr[17];
    cells.muB = ptr[18]; cells.nB = ptr[19]; cells.Vx = ptr[20]; cells.Vy = ptr[21]; cells.Vn = ptr[22];
    is3d_species sp{(int32_t)mcid.size(), mass.data(), sign.data(), deg.data(), bar.data()};
    is3d_grid grid{(int32_t)pT.size(), pT.data(), (int32_t)phi.size(), phi.data(), (int32_t)y.size(), y.data(),
                   (int32_t)eta.size(), eta.data(), etaw.data()};
    is3d_df_tables df{(int32_t)Tk.size(), Tk.data(), (int32_t)Bk.size(), Bk.data(), tab[0].data(), tab[1].data(), tab[2].data(),
                      tab[3].data(), tab[4].data(), tab[5].data(), tab[6].data(), tab[7].data(), tab[8].data(), tab[9].data()};
    is3d_options opts{};
    opts.dimension = dimension; opts.df_mode = df_mode; opts.include_baryon = include_baryon;
    opts.include_bulk_deltaf = include_bulk; opts.include_shear_deltaf = include_shear; opts.include_baryondiff_deltaf = include_diff;
    opts.regulate_deltaf = regulate; opts.outflow = outflow;
    opts.accumulate = 0; opts.device = rd.list.empty() ? -1 : rd.list[0]; opts.kernel_variant = variant;
    const int ny_eff = (dimension == 2) ? 1 : (int)y.size();
    std::vector<double> dN(mcid.size() * pT.size() * phi.size() * (size_t)ny_eff, 0.0);
    if (operation == 2) {
        // ---- emissionfunction.cpp:1522-1545: number of events, sampling, OSCAR list ----
        double oversample, min_num_hadrons, max_num_samples, sampler_seed, y_cut, fast, test_sampler, set_T, T_switch = 0.0;
        if (get_param("oversample", &oversample) || get_param("min_num_hadrons", &min_num_hadrons) || get_param("max_num_samples", &max_num_samples) ||
            get_param("sampler_seed", &sampler_seed) || get_param("y_cut", &y_cut) || get_param("fast", &fast) || get_param("test_sampler", &test_sampler) ||
            get_param("set_fo_temperature", &set_T))
            return IS3D_EINVAL;
        if ((int)set_T && get_param("t_switch", &T_switch)) return IS3D_EINVAL;
        is3d_sampler_test_bins bins{};
        if ((int)test_sampler) {                                                  // emissionfunction.cpp:205-222
            double yb, ec, eb, pl, pu, pb, t0b, t1b, tb, r0b, r1b, rb;
            if (get_param("y_bins", &yb) || get_param("eta_cut", &ec) || get_param("eta_bins", &eb) || get_param("pt_lower_cut", &pl) ||
                get_param("pt_upper_cut", &pu) || get_param("pt_bins", &pb) || get_param("tau_min", &t0b) || get_param("tau_max", &t1b) ||
                get_param("tau_bins", &tb) || get_param("r_min", &r0b) || get_param("r_max", &r1b) || get_param("r_bins", &rb))
                return IS3D_EINVAL;
            bins.y_cut = y_cut; bins.eta_cut = ec; bins.pT_lower_cut = pl; bins.pT_upper_cut = pu; bins.tau_min = t0b; bins.tau_max = t1b;
            bins.r_min = r0b; bins.r_max = r1b;
            bins.y_bins = (int)yb; bins.eta_bins = (int)eb; bins.pT_bins = (int)pb; bins.tau_bins = (int)tb; bins.r_bins = (int)rb;
        }
        // cell positions: columns 1, 2 of every supported surface format (readindata.cpp:343-346 etc.)
        std::vector<double> xs((size_t)n_cells, 0.0), ys((size_t)n_cells, 0.0);
        if (mem) {
            if (mem_x) xs.assign(mem_x, mem_x + n_cells);
            if (mem_y) ys.assign(mem_y, mem_y + n_cells);
        } else if (n_cells > 0) {   // they came with the surface (is3d_surface_arrays: arrays 23, 24)
            xs.assign(sa[23], sa[23] + n_cells);
            ys.assign(sa[24], sa[24] + n_cells);
        }
        int32_t n_alpha = 0, n_pts = 0;
        const char *gla_path = "tables/gla_roots_weights_32_points.txt";
        if (is3d_gla_read(gla_path, &n_alpha, &n_pts, nullptr, nullptr, 0)) DIE("%s", is3d_last_error());
        if (n_alpha < 3) DIE("%s: needs alpha = 0, 1, 2", gla_path);
        std::vector<double> groot((size_t)n_alpha * n_pts), gweight((size_t)n_alpha * n_pts);
        if (is3d_gla_read(gla_path, &n_alpha, &n_pts, groot.data(), gweight.data(), (int64_t)groot.size())) DIE("%s", is3d_last_error());
        is3d_sampler_inputs si{};
        si.n_events = 1; si.n_gla = n_pts;
        si.seed = sampler_seed < 0 ? (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() : (uint64_t)sampler_seed;   // :842-844
        si.y_cut = y_cut; si.first_cell = 0; si.x = xs.data(); si.y = ys.data();
        si.root1 = groot.data() + n_pts; si.weight1 = gweight.data() + n_pts;
        // df_mode 3 / 4 and fast mode: emissionfunction.cpp:1309-1321, sampling_kernels.cpp:852-869
        double T_avg_file = 0.0, E_avg_file = 0.0, P_avg_file = 0.0, muB_avg_file = 0.0;   // Plasma::load_thermodynamic_averages
        {
            FILE *tf = fopen("average_thermodynamic_quantities.dat", "r");
            if (!tf || fscanf(tf, "%lf %lf %lf %lf", &T_avg_file, &E_avg_file, &P_avg_file, &muB_avg_file) != 4) DIE("Error opening average thermodynamic file");
            fclose(tf);
        }
        is3d_feqmod_tables fqs{};
        if (feqmod || (int)fast) {
            double deta_min, mass_pion0;
            if (get_param("deta_min", &deta_min) || get_param("mass_pion0", &mass_pion0)) return IS3D_EINVAL;
            fqs.n_gla = n_pts;
            fqs.root1 = si.root1; fqs.weight1 = si.weight1;
            fqs.root2 = groot.data() + 2 * (size_t)n_pts; fqs.weight2 = gweight.data() + 2 * (size_t)n_pts;
            fqs.n_pdg = npdg; fqs.pdg_mass = pmass.data(); fqs.pdg_degeneracy = pg.data(); fqs.pdg_sign = ps.data();
            fqs.T_avg = T_avg_file; fqs.deta_min = deta_min; fqs.mass_pion0 = mass_pion0;
            si.feqmod = &fqs;
        }
        si.fast = (int)fast != 0;
        si.muB_avg = muB_avg_file;                                                // Plasma::baryon_chemical_potential, :858
        si.T_avg = T_avg_file;
        si.T_avg_switch = (int)set_T ? T_switch : T_avg_file;                     // :856
        if (si.fast) printf("Using fast mode: (Tavg, muBavg) = (%lf, %lf)\n", si.T_avg_switch, muB_avg_file);
        printf("iS3D Sampling Seed : %llu\n", (unsigned long long)si.seed);
        is3d_sampler_stats ss{};
        int64_t count = 0;
        double mean_yield = 0.0;   // calculate_total_yield's member mean_yield (:828), written by write_yield_list_toFile
        if ((int)oversample) {
            // emissionfunction.cpp:1524-1533: the analytic mean yield (calculate_total_yield) sizes the run
            double E_avg = 0.0, P_avg = 0.0, nB_avg = 0.0;
            {
                FILE *tf = fopen("average_thermodynamic_quantities.dat", "r");
                double t_, m_;
                if (!tf || fscanf(tf, "%lf %lf %lf %lf %lf", &t_, &E_avg, &P_avg, &m_, &nB_avg) != 5) DIE("Error opening average thermodynamic file");
                fclose(tf);
            }
            if (n_alpha < 4) DIE("%s: the yield estimate needs alpha = 0, 1, 2, 3", gla_path);
            is3d_feqmod_tables fqy = fqs;
            if (!si.feqmod) {   // the alpha = 2 nodes enter every df_mode's bulk density (J20)
                fqy.n_gla = n_pts;
                fqy.root1 = si.root1; fqy.weight1 = si.weight1;
                fqy.root2 = groot.data() + 2 * (size_t)n_pts; fqy.weight2 = gweight.data() + 2 * (size_t)n_pts;
            }
            is3d_sampler_inputs sy = si;
            sy.feqmod = &fqy;
            is3d_yield_inputs yi{T_avg_file, E_avg, P_avg, muB_avg_file, nB_avg, groot.data() + 3 * (size_t)n_pts, gweight.data() + 3 * (size_t)n_pts};
            double Ntotal = 0.0;
            printf("Total particle yield: ");
            int rc1 = is3d_total_yield(&cells, &sp, &df, &sy, &yi, &opts, &Ntotal, nullptr);
            if (rc1) DIE("is3d_total_yield failed (%d): %s", rc1, is3d_last_error());
            if (dimension == 2) printf("dN_dy ~ %lf\n\n", Ntotal / (2.0 * y_cut));
            printf("%lf\n", Ntotal);
            mean_yield = Ntotal;
            Ntotal = (double)fabsf((float)Ntotal);                                  // "prevent overflow", :1528
            // Nevents = min((int)ceil(MIN_NUM_HADRONS / Ntotal), MAX_NUM_SAMPLES); at least one event is sampled here
            si.n_events = (int32_t)std::max(1.0, std::min(std::ceil(min_num_hadrons / Ntotal), (double)(int)max_num_samples));
        }
        printf("Sampling %d event(s)\n", si.n_events);
        if (df_mode == 1) printf("Sampling particles with Grad 14 moment df...\n");                    // emissionfunction.cpp:1540-1541, :1602-1603
        if (df_mode == 2) printf("Sampling particles with Chapman Enskog df...\n");
        if (df_mode == 3) printf("Sampling particles with Mike's modified distribution...\n");
        if (df_mode == 4) printf("Sampling particles with Jonah's modified distribution...\n");
        int rc2 = is3d_sample_particles_multi(&cells, &sp, &df, &si, &opts, rd.list.empty() ? nullptr : rd.list.data(), (int32_t)rd.list.size(), nullptr, 0, &count, &ss);
        if (rc2) DIE("is3d_sample_particles failed (%d): %s", rc2, is3d_last_error());
        std::vector<is3d_particle> plist((size_t)std::max<int64_t>(count, 1));
        rc2 = is3d_sample_particles_multi(&cells, &sp, &df, &si, &opts, rd.list.empty() ? nullptr : rd.list.data(), (int32_t)rd.list.size(), plist.data(), count, &count, &ss);
        if (rc2) DIE("is3d_sample_particles failed (%d): %s", rc2, is3d_last_error());
        double t2s = now_s();
        printf("\nMomentum sampling efficiency = %f %%\n", 100.0 * (double)ss.n_acceptances / (double)std::max<int64_t>(ss.n_momentum_samples, 1));
        if (feqmod) printf("feqmod breaks down for %lld cells\n", (long long)ss.n_cells_breakdown);
        if ((int)test_sampler) {                                                  // emissionfunction.cpp:1545-1554
            printf("Writing the binned sampler test distributions...\n");
            if (is3d_write_sampler_tests("results", &bins, si.n_events, sp.n, mcid.data(), count, plist.data(), mean_yield)) DIE("%s", is3d_last_error());
        } else {
            printf("Writing sampled particles list to OSCAR File...\n");
            if (is3d_write_particle_list_osc("results/particle_list_osc.dat", si.n_events, count, plist.data(), mcid.data())) DIE("%s", is3d_last_error());
        }
        double t3s = now_s();
        printf("particles: %lld in %d event(s); hadrons drawn %lld; cells skipped (u.dsigma <= 0): %lld\n", (long long)count, si.n_events,
               (long long)ss.n_hadrons_drawn, (long long)ss.n_cells_skipped);
        printf("device time: prep %.3f ms, count %.3f ms, fill %.3f ms; h2d %.3f ms\n", ss.ms_prep, ss.ms_count, ss.ms_fill, ss.ms_h2d);
        printf("wall: read %.3f s, sampling %.3f s, write %.3f s\n", t1 - t0, t2s - t1, t3s - t2s);
        if (res) {
            // iS3D.cpp:178-184: the event lists go back to the caller (final_particles_)
            res->operation = 2; res->n_events = si.n_events; res->n_species = sp.n; res->n_particles = count;
            res->particles = (is3d_particle *)malloc(sizeof(is3d_particle) * (size_t)std::max<int64_t>(count, 1));
            res->mc_id = (int64_t *)malloc(sizeof(int64_t) * (size_t)sp.n);
            res->mass = (double *)malloc(sizeof(double) * (size_t)sp.n);
            if (!res->particles || !res->mc_id || !res->mass) return is3d::set_error(IS3D_ENOMEM, "out of memory for the particle list");
            memcpy(res->particles, plist.data(), sizeof(is3d_particle) * (size_t)count);
            memcpy(res->mc_id, mcid.data(), sizeof(int64_t) * (size_t)sp.n);
            memcpy(res->mass, mass.data(), sizeof(double) * (size_t)sp.n);
        }
        printf("Done sampling particles. Output stored in results folder. Goodbye!\n");
        return IS3D_OK;
    }
    is3d_status st{};
    int rc;
    if (vah) {
        // the call the reference has commented out (emissionfunction.cpp:1650-1654), with the coefficients the CUDA tree's reader
        // would have put into the surface (src/cuda/deltafReader.cu:224-278)
        printf("computing thermal spectra from vahydro (P_L matching) with df...\n");
        int32_t nL = 0, naL = 0;
        if (is3d_vah_df_read("deltaf_coefficients/vah", &nL, &naL, nullptr, nullptr, nullptr, 0)) DIE("%s", is3d_last_error());
        std::vector<double> Lg((size_t)nL), ag((size_t)naL), ct((size_t)5 * nL * naL);
        if (is3d_vah_df_read("deltaf_coefficients/vah", &nL, &naL, Lg.data(), ag.data(), ct.data(), (int64_t)ct.size())) DIE("%s", is3d_last_error());
        const size_t tn = (size_t)nL * naL;
        is3d_vah_df_tables vt{nL, naL, Lg.data(), ag.data(), ct.data(), ct.data() + tn, ct.data() + 2 * tn, ct.data() + 3 * tn, ct.data() + 4 * tn};
        is3d_vah_cells vc{};
        vc.n_cells = n_cells;
        const double **vf[25] = {&vc.tau, &vc.eta, &vc.ux, &vc.uy, &vc.un, &vc.dat, &vc.dax, &vc.day, &vc.dan, &vc.T, &vc.pitt, &vc.pitx, &vc.pity,
                                 &vc.pitn, &vc.pixx, &vc.pixy, &vc.pixn, &vc.piyy, &vc.piyn, &vc.pinn, &vc.bulkPi, &vc.Wx, &vc.Wy, &vc.Lambda, &vc.aL};
        for (int a = 0; a < 25; a++) *vf[a] = sa[a];
        rc = is3d_smooth_spectra_vah_df(&vc, &sp, &grid, &vt, &opts, dN.data(), &st);
    } else if (feqmod) {
        printf("computing thermal spectra from vhydro with feqmod...\n");
        // emissionfunction.cpp:1309-1319: Gauss-Laguerre tables, Plasma::load_thermodynamic_averages (the file written
        // above, read back as text), parameters deta_min and mass_pion0 (:184, :188)
        int32_t n_alpha = 0, n_pts = 0;
        const char *gla_path = "tables/gla_roots_weights_32_points.txt";
        if (is3d_gla_read(gla_path, &n_alpha, &n_pts, nullptr, nullptr, 0)) DIE("%s", is3d_last_error());
        if (n_alpha < 3) DIE("%s: needs alpha = 0, 1, 2", gla_path);
        std::vector<double> groot((size_t)n_alpha * n_pts), gweight((size_t)n_alpha * n_pts);
        if (is3d_gla_read(gla_path, &n_alpha, &n_pts, groot.data(), gweight.data(), (int64_t)groot.size())) DIE("%s", is3d_last_error());
        double deta_min, mass_pion0, T_avg = 0.0;
        if (get_param("deta_min", &deta_min) || get_param("mass_pion0", &mass_pion0)) return IS3D_EINVAL;
        {
            FILE *tf = fopen("average_thermodynamic_quantities.dat", "r");
            if (!tf || fscanf(tf, "%lf", &T_avg) != 1) DIE("Error opening average thermodynamic file");
            fclose(tf);
        }
        is3d_feqmod_tables fq{};
        fq.n_gla = n_pts;
        fq.root1 = groot.data() + n_pts; fq.weight1 = gweight.data() + n_pts;
        fq.root2 = groot.data() + 2 * (size_t)n_pts; fq.weight2 = gweight.data() + 2 * (size_t)n_pts;
        fq.n_pdg = npdg; fq.pdg_mass = pmass.data(); fq.pdg_degeneracy = pg.data(); fq.pdg_sign = ps.data();
        fq.T_avg = T_avg; fq.deta_min = deta_min; fq.mass_pion0 = mass_pion0;
        rc = is3d_smooth_spectra_multi(&cells, &sp, &grid, &df, &fq, &opts, rd.list.empty() ? nullptr : rd.list.data(), (int32_t)rd.list.size(),
                                       rd.reduce, dN.data(), &st, nullptr);
    } else {
        printf("computing thermal spectra from vhydro with df...\n");
        rc = is3d_smooth_spectra_multi(&cells, &sp, &grid, &df, nullptr, &opts, rd.list.empty() ? nullptr : rd.list.data(), (int32_t)rd.list.size(),
                                       rd.reduce, dN.data(), &st, nullptr);
    }
    if (!vah) {
        const int nd = rd.list.empty() ? is3d_device_count() : (int)rd.list.size();
        printf("devices: %d (cell-axis shards of ~%lld cells%s)\n", nd, (long long)((n_cells + nd - 1) / std::max(nd, 1)),
               nd > 1 ? (rd.reduce == IS3D_REDUCE_RCCL ? ", RCCL all-reduce of the spectrum" : ", shard-ordered device sum of the spectrum") : "");
    }
    if (rc) {
        const std::string msg = is3d_last_error();
        DIE("is3d_smooth_spectra failed (%d): %s", rc, msg.c_str());
    }
    if (feqmod && !vah) printf("\nfeqmod breaks down for %lld cells\n\n", (long long)st.n_cells_breakdown);   // smooth_kernels.cpp:989
    double t2 = now_s();
    if (is3d_write_results("results", dimension, sp.n, mcid.data(), grid.n_pT, pT.data(), pTw.data(), grid.n_phi, phi.data(),
                           phiw.data(), grid.n_y, y.data(), dN.data()))
        DIE("%s", is3d_last_error());
    double t3 = now_s();
    printf("species classes evaluated: %d of %d; cells skipped (u.dsigma <= 0): %lld\n", st.n_classes, sp.n, (long long)st.n_cells_skipped);
    printf("device time: prep %.3f ms, main %.3f ms (kernel variant %d), finalize %.3f ms; h2d %.3f ms, d2h %.3f ms\n", st.ms_prep,
           st.ms_main, st.kernel_variant, st.ms_finalize, st.ms_h2d, st.ms_d2h);
    const int src_kind = surf ? is3d_surface_source(surf) : -1;
    printf("wall: read %.3f s, spectra %.3f s, write %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
    if (src_kind >= 0)
        printf("surface: %s\n", src_kind == 2 ? "binary sidecar input/surface.dat.is3dcache (matches the text file; IS3D_NO_CACHE=1 to ignore)"
                                 : src_kind == 1 ? "text parsed, sidecar input/surface.dat.is3dcache written for the next run" : "text parsed");
    if (res) {
        res->operation = 1; res->n_species = sp.n; res->n_spectrum = (int64_t)dN.size();
        res->spectrum = (double *)malloc(sizeof(double) * dN.size());
        res->mc_id = (int64_t *)malloc(sizeof(int64_t) * (size_t)sp.n);
        res->mass = (double *)malloc(sizeof(double) * (size_t)sp.n);
        if (!res->spectrum || !res->mc_id || !res->mass) return is3d::set_error(IS3D_ENOMEM, "out of memory for the spectrum");
        memcpy(res->spectrum, dN.data(), sizeof(double) * dN.size());
        memcpy(res->mc_id, mcid.data(), sizeof(int64_t) * (size_t)sp.n);
        memcpy(res->mass, mass.data(), sizeof(double) * (size_t)sp.n);
    }
    printf("Done calculating particle spectra. Output stored in results folder. Goodbye!\n");
    return IS3D_OK;
}

extern "C" int is3d_run_particlization(const is3d_cells *surface, const double *x, const double *y, int32_t kernel_variant,
                                       is3d_run_result *result)
{
    return is3d_run_particlization_on(surface, x, y, kernel_variant, nullptr, 0, IS3D_REDUCE_ORDERED, result);
}

extern "C" int is3d_run_particlization_on(const is3d_cells *surface, const double *x, const double *y, int32_t kernel_variant,
                                          const int32_t *devices, int32_t n_devices, int32_t reduce, is3d_run_result *result)
{
    if (result) memset(result, 0, sizeof *result);
    RunDevices rd;
    if (int rc = parse_run_devices(devices, n_devices, reduce, rd)) return rc;
    return run_impl(surface, x, y, kernel_variant, rd, result);
}

extern "C" void is3d_run_result_free(is3d_run_result *r)
{
    if (!r) return;
    free(r->particles); free(r->mc_id); free(r->mass); free(r->spectrum);
    memset(r, 0, sizeof *r);
}
