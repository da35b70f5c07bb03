#include "icpx_io.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

namespace icp {

static size_t ply_type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}

long loadFile(const std::string& file_name, PointCloudT& cloud, std::string* err) {
    auto fail = [&](const std::string& m) { if (err) *err = file_name + ": " + m; return -1L; };
    std::ifstream f(file_name, std::ios::binary);
    if (!f) return fail("cannot open");
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") return fail("not a PLY file");
    bool binary = false, ascii = false, in_vertex = false, seen_vertex = false;
    long nvert = -1;
    size_t stride = 0;
    std::vector<std::pair<std::string, std::string>> props;   // (type, name) of the vertex element
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream is(line);
        std::string tok;
        is >> tok;
        if (tok == "format") { std::string fmt; is >> fmt; binary = fmt == "binary_little_endian"; ascii = fmt == "ascii"; }
        else if (tok == "element") {
            std::string name; long n; is >> name >> n;
            if (name == "vertex") {
                if (seen_vertex) return fail("two vertex elements");
                in_vertex = true; seen_vertex = true; nvert = n;
            } else {
                if (!seen_vertex) return fail("an element precedes the vertex element");
                in_vertex = false;
            }
        } else if (tok == "property" && in_vertex) {
            std::string type, name; is >> type >> name;
            if (type == "list") return fail("list property in the vertex element");
            const size_t sz = ply_type_size(type);
            if (!sz) return fail("unknown property type " + type);
            props.emplace_back(type, name); stride += sz;
        } else if (tok == "end_header") break;
    }
    if (!binary && !ascii) return fail("unsupported PLY format (need ascii or binary_little_endian)");
    if (nvert < 0 || props.size() < 3) return fail("no vertex element with x, y, z");
    for (int c = 0; c < 3; c++) {
        const char* want[3] = {"x", "y", "z"};
        if (props[c].second != want[c] || ply_type_size(props[c].first) != 4 || props[c].first.find("int") != std::string::npos)
            return fail("x, y, z must be the first three float properties of a vertex");
    }
    cloud.points.clear();
    cloud.points.reserve((size_t)nvert);
    if (binary) {
        std::vector<char> buf((size_t)nvert * stride);
        f.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) return fail("vertex data truncated");
        for (long i = 0; i < nvert; i++) {
            float v[3];
            std::memcpy(v, buf.data() + (size_t)i * stride, 12);
            if (std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2])) cloud.points.emplace_back(v[0], v[1], v[2]);
        }
    } else {
        for (long i = 0; i < nvert; i++) {
            if (!std::getline(f, line)) return fail("vertex data truncated");
            std::istringstream is(line);
            float v[3];
            if (!(is >> v[0] >> v[1] >> v[2])) return fail("bad vertex line");
            if (std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2])) cloud.points.emplace_back(v[0], v[1], v[2]);
        }
    }
    return (long)cloud.points.size();
}

double normalize_goicp(PointCloudT& model, PointCloudT& data, double cm[3], double cd[3]) {
    auto centroid = [](const PointCloudT& c, double m[3]) {
        m[0] = m[1] = m[2] = 0;
        for (const Point3f& p : c.points) { m[0] += p.x; m[1] += p.y; m[2] += p.z; }
        for (int k = 0; k < 3; k++) m[k] /= (double)std::max<size_t>(1, c.points.size());
    };
    double mm[3], md[3];
    centroid(model, mm); centroid(data, md);
    double ext = 0;
    for (const Point3f& p : model.points) ext = std::max({ext, std::fabs(p.x - mm[0]), std::fabs(p.y - mm[1]), std::fabs(p.z - mm[2])});
    for (const Point3f& p : data.points) ext = std::max({ext, std::fabs(p.x - md[0]), std::fabs(p.y - md[1]), std::fabs(p.z - md[2])});
    const double s = ext > 0 ? 1.0 / ext : 1.0;
    for (Point3f& p : model.points) { p.x = (float)((p.x - mm[0]) * s); p.y = (float)((p.y - mm[1]) * s); p.z = (float)((p.z - mm[2]) * s); }
    for (Point3f& p : data.points) { p.x = (float)((p.x - md[0]) * s); p.y = (float)((p.y - md[1]) * s); p.z = (float)((p.z - md[2]) * s); }
    if (cm) std::memcpy(cm, mm, sizeof mm);
    if (cd) std::memcpy(cd, md, sizeof md);
    return s;
}

}  // namespace icp
