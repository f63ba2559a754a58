// Shadows ONE reference header: voxblox/interpolator/interpolator.h, which mesh/mesh_integrator.h includes and does
// not use (MeshIntegrator reads voxels directly); the real one needs a large part of Eigen's array API.  Nothing that
// is compared with the restatement goes through it.  TEST INFRASTRUCTURE ONLY.
#pragma once
