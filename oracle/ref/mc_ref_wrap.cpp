// The marching-cubes tables as the reference's MarchingCubes.cpp defines them (see mc_stub/).  TEST INFRASTRUCTURE ONLY.
#include <open_chisel/marching_cubes/MarchingCubes.h>

extern "C" void ref_chisel_mc_tables(int* triangle_table, int* edge_index_pairs) {
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 16; ++j) triangle_table[16 * i + j] = chisel::MarchingCubes::triangleTable[i][j];
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 2; ++j) edge_index_pairs[2 * i + j] = chisel::MarchingCubes::edgeIndexPairs[i][j];
}
