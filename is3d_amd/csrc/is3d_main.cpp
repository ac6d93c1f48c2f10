// is3d_main.cpp -- `iS3D`-compatible command line driver: run it inside an iS3D run directory (iS3D_parameters.dat, input/,
// PDG/, tables/, deltaf_coefficients/, results/).  Everything happens in is3d_run_particlization (is3d_run.cpp), the
// counterpart of RuniS3D.cpp -> IS3D::run_particlization(1) (/root/reference/src/cpp/RuniS3D.cpp, iS3D.cpp:74-192).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/is3d_amd.h"

int main(int argc, char **argv)
{
    int variant = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--variant") && i + 1 < argc) variant = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--help")) {
            printf("usage: %s [--variant 1|2|3|4]   (run inside an iS3D run directory)\n", argv[0]);
            return 0;
        }
    }
    return is3d_run_particlization(NULL, NULL, NULL, variant, NULL) == IS3D_OK ? 0 : 1;
}
