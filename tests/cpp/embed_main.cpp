// Test host for include/iS3D_amd.hpp: does what a framework embedding iS3D does (src/cpp/iS3D.h:60-96) -- fills the surface
// vectors, calls run_particlization(0) in a run directory, reads final_particles_ -- and prints the list for the Python test.
// usage: embed_main surface21.txt   (columns: tau x y eta dat dax day dan E T P ux uy un pixx pixy pixn piyy piyn pinn Pi)
#include <cstdio>
#include <fstream>
#include <vector>

#include "iS3D_amd.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1]);
    std::vector<std::vector<double> > col(21);
    double v;
    size_t k = 0;
    while (f >> v) { col[k % 21].push_back(v); k++; }
    is3d_amd::IS3D is3d;
    is3d.read_fo_surf_from_memory(col[0], col[1], col[2], col[3], col[4], col[5], col[6], col[7], col[8], col[9], col[10], col[11], col[12],
                                  col[13], col[14], col[15], col[16], col[17], col[18], col[19], col[20]);
    try {
        is3d.run_particlization(0);
    } catch (const std::exception &e) {
        fprintf(stderr, "embed_main: %s\n", e.what());
        return 1;
    }
    printf("EVENTS %zu SPECTRUM %zu\n", is3d.final_particles_.size(), is3d.dN_pTdpTdphidy_.size());
    for (size_t ev = 0; ev < is3d.final_particles_.size(); ev++)
        for (const is3d_amd::Sampled_Particle &p : is3d.final_particles_[ev])
            printf("P %zu %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", ev, p.chosen_index, p.mcID, p.mass, p.tau, p.x,
                   p.y, p.eta, p.t, p.z, p.E, p.px, p.py, p.pz);
    if (!is3d.dN_pTdpTdphidy_.empty()) {
        double s = 0.0;
        for (double d : is3d.dN_pTdpTdphidy_) s += d;
        printf("SUM %.17g FIRST %.17g\n", s, is3d.dN_pTdpTdphidy_[0]);
    }
    return 0;
}
