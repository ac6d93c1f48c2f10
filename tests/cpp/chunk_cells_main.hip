// Host-only check of is3d::chunk_cells (is3d_amd/csrc/cf_device.h): prints "chunk c0 c1" for every chunk of a partition.
// usage: chunk_cells_main n_cells nch nch_small
#include <cstdio>
#include <cstdlib>
#include "../../is3d_amd/csrc/cf_device.h"

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    is3d::MainGeom g{};
    g.n_cells = atoi(argv[1]); g.nch = atoi(argv[2]); g.nch_small = atoi(argv[3]);
    for (int c = 0; c < g.nch; c++) {
        int c0, c1;
        is3d::chunk_cells(g, c, c0, c1);
        printf("%d %d %d\n", c, c0, c1);
    }
    return 0;
}
