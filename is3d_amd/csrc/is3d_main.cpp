// is3d_main.cpp -- `iS3D`-compatible command line driver: run it inside an iS3D run directory (iS3D_parameters.dat, input/,
// PDG/, tables/, deltaf_coefficients/, results/).  Everything happens in is3d_run_particlization (is3d_run.cpp), the
// counterpart of RuniS3D.cpp -> IS3D::run_particlization(1) (/root/reference/src/cpp/RuniS3D.cpp, iS3D.cpp:74-192).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <vector>

#include "../../include/is3d_amd.h"

int main(int argc, char **argv)
{
    int variant = 0, reduce = -1;
    std::vector<int32_t> devices;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--variant") && i + 1 < argc) variant = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            // "0,1,2": one cell-axis shard per entry (an ordinal may repeat); default: IS3D_DEVICES, else every visible device
            for (const char *p = argv[++i]; *p;) {
                char *end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p || v < 0 || (*end && *end != ',')) { fprintf(stderr, "--devices: a comma separated list of HIP device ordinals\n"); return 2; }
                devices.push_back((int32_t)v);
                p = *end ? end + 1 : end;
            }
        } else if (!strcmp(argv[i], "--reduce") && i + 1 < argc) {
            ++i;
            if (!strcmp(argv[i], "rccl")) reduce = IS3D_REDUCE_RCCL;
            else if (!strcmp(argv[i], "ordered")) reduce = IS3D_REDUCE_ORDERED;
            else { fprintf(stderr, "--reduce ordered|rccl\n"); return 2; }
        } else if (!strcmp(argv[i], "--help")) {
            printf("usage: %s [--variant 1|2|3|4] [--devices 0,1,...] [--reduce ordered|rccl]   (run inside an iS3D run directory)\n"
                   "  operation = 1 shards the freezeout cells over the devices (default: every visible GPU; IS3D_DEVICES, IS3D_REDUCE)\n", argv[0]);
            return 0;
        }
    }
    if (devices.empty() && reduce < 0) return is3d_run_particlization(NULL, NULL, NULL, variant, NULL) == IS3D_OK ? 0 : 1;
    if (devices.empty()) {   // --reduce alone: every visible device
        for (int d = 0; d < is3d_device_count(); d++) devices.push_back(d);
        if (devices.empty()) devices.push_back(0);
    }
    return is3d_run_particlization_on(NULL, NULL, NULL, variant, devices.data(), (int32_t)devices.size(),
                                      reduce < 0 ? IS3D_REDUCE_ORDERED : reduce, NULL) == IS3D_OK ? 0 : 1;
}
