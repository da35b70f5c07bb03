// Cloud input and normalisation for the command-line driver: what src/main.cpp does through PCL.
#pragma once
#include <string>
#include "icpx_host.hpp"

namespace icp {

// Reads the vertex positions of a PLY file (ascii or binary_little_endian; x, y, z must be the first
// three float properties of the vertex element, as in every scan the reference ships) and drops vertices
// with a non-finite coordinate. Stands in for loadFile(): pcl::io::loadPolygonFile +
// removeNaNFromPointCloud (src/main.cpp:35-51). Returns the number of points, or -1 with `err` set.
long loadFile(const std::string& file_name, PointCloudT& cloud, std::string* err = nullptr);

// Isotropic Go-ICP normalisation: each cloud is moved to its own centroid and both are multiplied by one
// scale so that every coordinate lies in [-1, 1] (the translation domain of GoIcp is [-0.5, 0.5]^3,
// src/GoIcp.cpp:92-96). The reference's own _normal() (src/main.cpp:166-207) is per-axis and unused.
// Returns the scale; the centroids are written to cm / cd when given.
double normalize_goicp(PointCloudT& model, PointCloudT& data, double cm[3] = nullptr, double cd[3] = nullptr);

}  // namespace icp
