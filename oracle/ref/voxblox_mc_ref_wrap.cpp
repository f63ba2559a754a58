// C entry points over voxblox's OWN marching cubes — mesh/marching_cubes.h (meshCube, interpolateEdgeVertices,
// interpolateVertex, calculateVertexConfiguration) and src/mesh/marching_cubes.cc (the triangle table and the edge
// pairs) — compiled where they lie under /root/reference against the stand-ins of oracle/ref/vbx_shim
// -> oracle/_ref/libvoxblox_ref.so.  tests/test_oracle_pinned.py checks vb_mesh_cube and the tables of
// oracle/tsdf_voxblox.c (and the product's copy of the table) against these.
// TEST INFRASTRUCTURE ONLY: nothing under plvs_amd/ may call into this file.
#include <cstring>

#include "voxblox/mesh/marching_cubes.h"

extern "C" {

void ref_voxblox_mc_tables(int* triangle_table, int* edge_index_pairs) {
  std::memcpy(triangle_table, voxblox::MarchingCubes::kTriangleTable, sizeof(int) * 256 * 16);
  std::memcpy(edge_index_pairs, voxblox::MarchingCubes::kEdgeIndexPairs, sizeof(int) * 12 * 2);
}

// MarchingCubes::meshCube(vertex_coords, vertex_sdf, &next_index, &mesh) on one cube: corner_coords 8 x 3
// (corner-major), corner_sdf 8.  Returns the number of vertices; vertices / normals hold up to 15 x 3 floats.
int ref_voxblox_mesh_cube(const float* corner_coords, const float* corner_sdf, float* vertices, float* normals) {
  Eigen::Matrix<voxblox::FloatingPoint, 3, 8> coords;
  Eigen::Matrix<voxblox::FloatingPoint, 8, 1> sdf;
  for (int i = 0; i < 8; ++i) {
    coords.col(i) = voxblox::Point(corner_coords[3 * i], corner_coords[3 * i + 1], corner_coords[3 * i + 2]);
    sdf(i) = corner_sdf[i];
  }
  voxblox::Mesh mesh;
  voxblox::VertexIndex next_index = 0;
  voxblox::MarchingCubes::meshCube(coords, sdf, &next_index, &mesh);
  const int n = (int)mesh.vertices.size();
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      vertices[3 * i + k] = mesh.vertices[i][k];
      normals[3 * i + k] = mesh.normals[i][k];
    }
  return n;
}

}  // extern "C"
