// TEST INFRASTRUCTURE ONLY — thin extern "C" harness over the parts of the reference that build
// from their own standard-C++ sources, compiled where they lie under /root/reference/src (see
// oracle/Makefile; output goes to oracle/_ref/, which is git-ignored). No reference source is
// copied: this file only #includes / links
//   src/jly_icp3d.cpp  (ICP3D<T>, its nanoflann PointCloud adaptor; pulls in src/nanoflann.hpp, src/matrix.h)
//   src/matrix.cpp     (Matrix::svd)
//   src/jly_3ddt.cpp   (DT3D::Build / DT3D::Distance)
// and is used by tests/test_oracle_ref.py to pin oracle/oracle.cpp against the reference's own code.
#include <cstdint>
#include <cstring>
#include <vector>

#include "jly_icp3d.cpp"
#include "jly_3ddt.h"

extern "C" {

// nanoflann 1-NN exactly as ICP3D uses it (src/jly_icp3d.cpp:147-153 build, :229 query), T = double.
void ref_nanoflann_nn(const double* tgt, int64_t m, const double* q, int64_t n, int64_t* idx, double* d2) {
    PointCloud<double> pc;
    pc.pts.resize(m);
    for (int64_t i = 0; i < m; i++) { pc.pts[i].x = tgt[3 * i]; pc.pts[i].y = tgt[3 * i + 1]; pc.pts[i].z = tgt[3 * i + 2]; }
    typedef KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<double, PointCloud<double> >, PointCloud<double>, 3> Tree;
    Tree tree(3, pc, KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    for (int64_t i = 0; i < n; i++) {
        size_t ri = 0; double rd = 0;
        tree.knnSearch(q + 3 * i, 1, &ri, &rd);
        idx[i] = (int64_t)ri; d2[i] = rd;
    }
}

// nanoflann knnSearch with k results per query (ascending squared distances).
void ref_nanoflann_knn(const double* tgt, int64_t m, const double* q, int64_t n, int k, int64_t* idx, double* d2) {
    PointCloud<double> pc;
    pc.pts.resize(m);
    for (int64_t i = 0; i < m; i++) { pc.pts[i].x = tgt[3 * i]; pc.pts[i].y = tgt[3 * i + 1]; pc.pts[i].z = tgt[3 * i + 2]; }
    typedef KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<double, PointCloud<double> >, PointCloud<double>, 3> Tree;
    Tree tree(3, pc, KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    std::vector<size_t> ri(k);
    for (int64_t i = 0; i < n; i++) {
        tree.knnSearch(q + 3 * i, (size_t)k, ri.data(), d2 + i * k);
        for (int s = 0; s < k; s++) idx[i * k + s] = (int64_t)ri[s];
    }
}

// Matrix::svd (src/matrix.cpp:582-810): A = U diag(W) V^T, all 3x3 row-major.
void ref_matrix_svd3(const double* A9, double* U9, double* W3, double* V9) {
    Matrix A(3, 3, A9), U, W, V;
    A.svd(U, W, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U9[3 * i + j] = U.val[i][j]; V9[3 * i + j] = V.val[i][j]; }
    for (int i = 0; i < 3; i++) W3[i] = W.val[i][0];
}

// ICP3D<double>::Run (src/jly_icp3d.cpp:181-295) with a fixed iteration count and no early stop
// (err_diff = -1 can never satisfy the break test at :257). R9 / t3 in-out, row-major.
double ref_icp3d_run(const double* model, int64_t m, const double* data, int64_t n, double* R9, double* t3, int64_t iters) {
    ICP3D<double> icp;
    icp.do_trim = false;
    icp.Build(const_cast<double*>(model), (size_t)m);
    Matrix R(3, 3, R9), t(3, 1, t3);
    double err = icp.Run(const_cast<double*>(data), (size_t)n, R, t, (size_t)iters, -1.0);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R9[3 * i + j] = R.val[i][j]; t3[i] = t.val[i][0]; }
    return err;
}

// ICP3D<double>::Run with the caller's trimming and stop settings (public fields do_trim, trim_fraction; arguments
// max_iter, err_diff).
double ref_icp3d_run_ex(const double* model, int64_t m, const double* data, int64_t n, double* R9, double* t3, int64_t max_iter,
                        double err_diff, int do_trim, double trim_fraction) {
    ICP3D<double> icp;
    icp.do_trim = do_trim != 0;
    icp.trim_fraction = trim_fraction;
    icp.Build(const_cast<double*>(model), (size_t)m);
    Matrix R(3, 3, R9), t(3, 1, t3);
    double err = icp.Run(const_cast<double*>(data), (size_t)n, R, t, (size_t)max_iter, err_diff);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R9[3 * i + j] = R.val[i][j]; t3[i] = t.val[i][0]; }
    return err;
}

// DT3D (src/jly_3ddt.cpp:731-856). expandFactor is left uninitialised by the constructor (:727-729);
// the harness sets it. Returns an opaque handle.
void* ref_dt3d_build(const double* x, const double* y, const double* z, int n, double expandFactor) {
    DT3D* dt = new DT3D;
    dt->expandFactor = expandFactor;
    dt->Build(const_cast<double*>(x), const_cast<double*>(y), const_cast<double*>(z), n);
    return dt;
}
void ref_dt3d_info(void* h, double* out8) {
    DT3D* dt = (DT3D*)h;
    out8[0] = dt->SIZE; out8[1] = dt->scale; out8[2] = dt->xMin; out8[3] = dt->yMin; out8[4] = dt->zMin;
    out8[5] = dt->xMax; out8[6] = dt->yMax; out8[7] = dt->zMax;
}
void ref_dt3d_distance(void* h, const double* q, int64_t n, float* out) {
    DT3D* dt = (DT3D*)h;
    for (int64_t i = 0; i < n; i++) out[i] = dt->Distance(q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}
void ref_dt3d_free(void* h) { delete (DT3D*)h; }

}  // extern "C"
