// iS3D_amd.hpp -- the reference's embedding API on top of the C ABI (is3d_amd.h): class IS3D with the member names,
// method names and argument order of /root/reference/src/cpp/iS3D.h:19-96, and Sampled_Particle of src/cpp/particle.h:36-62,
// so that a host framework that embeds iS3D (fills the surface vectors, calls run_particlization(0), reads
// final_particles_) switches by changing the include and the link line (-lis3d_amd).  Header-only; C++11.
//
// Differences from the reference class:
//   - lives in namespace is3d_amd;
//   - errors throw std::runtime_error(is3d_last_error()) instead of printf + exit(-1);
//   - operation = 1 additionally leaves the spectrum in dN_pTdpTdphidy_ (the reference only writes files);
//   - pinn is accepted and ignored, as in the reference (iS3D.cpp:100-134 never copies it; the kernels reconstruct it).
#pragma once
#include <stdexcept>
#include <vector>

#include "is3d_amd.h"

namespace is3d_amd {

class Sampled_Particle {   // src/cpp/particle.h:36-62
public:
    int chosen_index = 0;  // index into PDG/chosen_particles.dat
    int mcID = 0;
    double mass = 0.0;
    double tau = 0.0, x = 0.0, y = 0.0, eta = 0.0;
    double t = 0.0, z = 0.0;
    double E = 0.0, px = 0.0, py = 0.0, pz = 0.0;
};

class IS3D {
public:
    // the freezeout surface (src/cpp/iS3D.h:28-58): Milne coordinates, covariant dsigma_mu, contravariant u^mu and pi^{mu nu};
    // E, T, P, pi, Pi in GeV and GeV/fm^3 (no hbar*c conversion happens on this path, iS3D.cpp:107-131)
    std::vector<double> tau, x, y, eta;
    std::vector<double> dsigma_tau, dsigma_x, dsigma_y, dsigma_eta;
    std::vector<double> E, T, P;
    std::vector<double> ux, uy, un;
    std::vector<double> pixx, pixy, pixn, piyy, piyn, pinn;
    std::vector<double> Pi;

    std::vector<std::vector<Sampled_Particle> > final_particles_;   // one list per event (operation = 2)
    std::vector<double> dN_pTdpTdphidy_;                             // operation = 1, species fastest (extension)
    int kernel_variant = 0;                                          // 0 = library default

    void read_fo_surf_from_file() {}                                 // declared but never defined in the reference

    void read_fo_surf_from_memory(std::vector<double> tau_in, std::vector<double> x_in, std::vector<double> y_in,
                                  std::vector<double> eta_in, std::vector<double> dsigma_tau_in, std::vector<double> dsigma_x_in,
                                  std::vector<double> dsigma_y_in, std::vector<double> dsigma_eta_in, std::vector<double> E_in,
                                  std::vector<double> T_in, std::vector<double> P_in, std::vector<double> ux_in,
                                  std::vector<double> uy_in, std::vector<double> un_in, std::vector<double> pixx_in,
                                  std::vector<double> pixy_in, std::vector<double> pixn_in, std::vector<double> piyy_in,
                                  std::vector<double> piyn_in, std::vector<double> pinn_in, std::vector<double> Pi_in)
    {
        tau = tau_in; x = x_in; y = y_in; eta = eta_in;
        dsigma_tau = dsigma_tau_in; dsigma_x = dsigma_x_in; dsigma_y = dsigma_y_in; dsigma_eta = dsigma_eta_in;
        E = E_in; T = T_in; P = P_in;
        ux = ux_in; uy = uy_in; un = un_in;
        pixx = pixx_in; pixy = pixy_in; pixn = pixn_in; piyy = piyy_in; piyn = piyn_in; pinn = pinn_in;
        Pi = Pi_in;
    }

    // fo_from_file = 1: read input/surface.dat; 0: use the vectors above (src/cpp/iS3D.cpp:74-192)
    void run_particlization(int fo_from_file)
    {
        is3d_run_result res;
        int rc;
        if (fo_from_file) {
            rc = is3d_run_particlization(NULL, NULL, NULL, kernel_variant, &res);
        } else {
            const size_t n = tau.size();
            const std::vector<double> *all[] = {&x, &y, &eta, &dsigma_tau, &dsigma_x, &dsigma_y, &dsigma_eta, &E, &T, &P, &ux, &uy, &un,
                                                &pixx, &pixy, &pixn, &piyy, &piyn, &Pi};
            for (size_t i = 0; i < sizeof all / sizeof all[0]; i++)
                if (all[i]->size() != n) throw std::runtime_error("IS3D: the surface vectors have different lengths");
            is3d_cells c = is3d_cells();
            c.n_cells = (int64_t)n;
            c.tau = tau.data(); c.eta = eta.data();
            c.dat = dsigma_tau.data(); c.dax = dsigma_x.data(); c.day = dsigma_y.data(); c.dan = dsigma_eta.data();
            c.ux = ux.data(); c.uy = uy.data(); c.un = un.data();
            c.T = T.data(); c.P = P.data(); c.E = E.data();
            c.pixx = pixx.data(); c.pixy = pixy.data(); c.pixn = pixn.data(); c.piyy = piyy.data(); c.piyn = piyn.data();
            c.bulkPi = Pi.data();
            rc = is3d_run_particlization(&c, x.data(), y.data(), kernel_variant, &res);
        }
        if (rc != IS3D_OK) {
            is3d_run_result_free(&res);
            throw std::runtime_error(is3d_last_error());
        }
        final_particles_.clear();
        dN_pTdpTdphidy_.clear();
        if (res.operation == 2) {   // iS3D.cpp:178-184
            final_particles_.resize((size_t)res.n_events);
            for (int64_t i = 0; i < res.n_particles; i++) {
                const is3d_particle &q = res.particles[i];
                Sampled_Particle s;
                s.chosen_index = q.species; s.mcID = (int)res.mc_id[q.species]; s.mass = res.mass[q.species];
                s.tau = q.tau; s.x = q.x; s.y = q.y; s.eta = q.eta; s.t = q.t; s.z = q.z;
                s.E = q.E; s.px = q.px; s.py = q.py; s.pz = q.pz;
                final_particles_[(size_t)q.event].push_back(s);
            }
        } else if (res.spectrum) {
            dN_pTdpTdphidy_.assign(res.spectrum, res.spectrum + res.n_spectrum);
        }
        is3d_run_result_free(&res);
    }
};

}  // namespace is3d_amd
