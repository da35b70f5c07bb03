// TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference's ICP / Go-ICP
// correspondence hot path. Nothing in the product path (icp_amd/, include/) may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// Pinning status: the reference ships no golden vectors or tests (src/main.cpp:54-243 are visual /
// timing drivers), and its live translation units (src/kdtree.cpp, src/SingleThreadIcp.cpp,
// src/testLDT.cpp, src/GoIcp.cpp) need PCL and boost headers that this image lacks, so they are
// unbuildable here. The pieces of the reference that are plain standard C++ (src/nanoflann.hpp,
// src/matrix.cpp, src/jly_3ddt.cpp) are compiled from where they lie into oracle/_ref/ (see
// oracle/Makefile) and this file is checked against them in tests/test_oracle_ref.py:
//   * nn_brute / kd-tree distances  == nanoflann knnSearch distances (bit-equal doubles)
//   * svd3 / kabsch                 == Matrix::svd of src/matrix.cpp (1e-12)
//   * exact EDT                     ~= DT3D chamfer DT of src/jly_3ddt.cpp (within its approximation error)
// The ICP trajectory (SingleThreadIcp) and the BnB logic have no runnable reference here; they are
// pinned only by figures recorded from runs of the reference before this code was written (tests/test_oracle_cpu.py) —
// otherwise "parity unpinned".
//
// Every function cites the reference lines it restates. Written from scratch; no reference code is
// copied. Build: see oracle/Makefile (g++ -O2 -fopenmp -ffp-contract=off).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <queue>
#include <vector>
#include <omp.h>

namespace {

// ---------------------------------------------------------------------------------------------
// Squared distance exactly as the reference evaluates it: doubles, x then y then z, no fused
// multiply-add (src/kdtree.cpp:131-133 accumulates pow(d,2) into 0.0 in dimension order).
// ---------------------------------------------------------------------------------------------
inline double sqdist(const double* p, const double* q) {
    double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    double acc = 0.0;
    acc += dx * dx;
    acc += dy * dy;
    acc += dz * dz;
    return acc;
}

// ---------------------------------------------------------------------------------------------
// kd-tree: implicit-heap layout, median split by std::nth_element cycling dimensions, recursive
// single-nearest search that keeps the FIRST visited candidate under strict '<'.
// Restates src/kdtree.cpp:58-89 (build), :107-150 (get_closest / k_close), :91-105 (Distance).
// ---------------------------------------------------------------------------------------------
struct KdNode {
    double f[3];
    int index;
    int dim;
};

struct KdTree {
    std::vector<KdNode> work;   // Node[] in the reference (points being partitioned)
    std::vector<KdNode> heap;   // data[] in the reference, 1-based implicit heap
    std::vector<uint8_t> used;  // flag[] in the reference
    int64_t n = 0;

    void build(int64_t o, int64_t l, int64_t r, int dep) {
        if (l > r) return;
        const int d = dep % 3;
        const int64_t lc = o << 1, rc = lc | 1;
        if ((size_t)rc >= used.size()) { used.resize(rc + 1, 0); heap.resize(rc + 1); }
        used[o] = 1; used[lc] = 0; used[rc] = 0;
        const int64_t mid = (l + r) >> 1;
        std::nth_element(work.begin() + l, work.begin() + mid, work.begin() + r + 1,
                         [d](const KdNode& a, const KdNode& b) { return a.f[d] < b.f[d]; });
        heap[o] = work[mid];
        heap[o].dim = d;
        build(lc, l, mid - 1, dep + 1);
        build(rc, mid + 1, r, dep + 1);
    }

    // k_close as written: visit the near child, then this node (push while the queue is short, else replace the
    // largest if strictly smaller), then the far child if the splitting plane is closer than the largest kept
    // (always, while the queue is short).
    void search_k(const double* p, int64_t o, int k, std::priority_queue<double>& Q) const {
        if ((size_t)o >= used.size() || !used[o]) return;
        const KdNode& nd = heap[o];
        int64_t lc = o << 1, rc = lc | 1;
        if (p[nd.dim] > nd.f[nd.dim]) std::swap(lc, rc);
        const double cur = sqdist(p, nd.f);
        if ((size_t)lc < used.size() && used[lc]) search_k(p, lc, k, Q);
        bool go;
        if ((int)Q.size() < k) { Q.push(cur); go = true; }
        else {
            if (cur < Q.top()) { Q.pop(); Q.push(cur); }
            const double dd = p[nd.dim] - nd.f[nd.dim];
            go = dd * dd < Q.top();
        }
        if (go && (size_t)rc < used.size() && used[rc]) search_k(p, rc, k, Q);
    }

    struct Best { double d2; int index; bool have; };

    void search(const double* p, int64_t o, Best& b) const {
        if ((size_t)o >= used.size() || !used[o]) return;
        const KdNode& nd = heap[o];
        int64_t lc = o << 1, rc = lc | 1;
        if (p[nd.dim] > nd.f[nd.dim]) std::swap(lc, rc);
        const double cur = sqdist(p, nd.f);
        if ((size_t)lc < used.size() && used[lc]) search(p, lc, b);
        bool go;
        if (!b.have) { b.have = true; b.d2 = cur; b.index = nd.index; go = true; }
        else {
            if (cur < b.d2) { b.d2 = cur; b.index = nd.index; }
            const double dd = p[nd.dim] - nd.f[nd.dim];
            go = dd * dd < b.d2;
        }
        if (go && (size_t)rc < used.size() && used[rc]) search(p, rc, b);
    }
};

// ---------------------------------------------------------------------------------------------
// 3x3 SVD by cyclic two-sided Jacobi on A^T A (eigenvectors -> V), U from A V / sigma with
// Gram-Schmidt completion. Only R = V U^T (after the reflection fix) is consumed, which is unique
// for non-degenerate input; replaces Eigen::JacobiSVD at src/SingleThreadIcp.cpp:252.
// ---------------------------------------------------------------------------------------------
void jacobi_eig3(double S[3][3], double V[3][3]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
        double diag = std::fabs(S[0][0]) + std::fabs(S[1][1]) + std::fabs(S[2][2]);
        if (off <= 1e-300 || off <= 1e-32 * diag) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (S[p][q] == 0.0) continue;
            double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
            double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) {  // S <- S J
                double a = S[k][p], b = S[k][q];
                S[k][p] = c * a - s * b; S[k][q] = s * a + c * b;
            }
            for (int k = 0; k < 3; k++) {  // S <- J^T S
                double a = S[p][k], b = S[q][k];
                S[p][k] = c * a - s * b; S[q][k] = s * a + c * b;
            }
            for (int k = 0; k < 3; k++) {
                double a = V[k][p], b = V[k][q];
                V[k][p] = c * a - s * b; V[k][q] = s * a + c * b;
            }
        }
    }
}

void svd3(const double A[3][3], double U[3][3], double sig[3], double V[3][3]) {
    double S[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double a = 0; for (int k = 0; k < 3; k++) a += A[k][i] * A[k][j];
        S[i][j] = a;
    }
    double W[3][3];
    jacobi_eig3(S, W);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return S[a][a] > S[b][b]; });
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 3; i++) V[i][j] = W[i][ord[j]];
        sig[j] = std::sqrt(std::max(S[ord[j]][ord[j]], 0.0));
    }
    // U columns = A v_j / sigma_j, orthonormalised; the last column is completed by a cross product
    // so that a rank-2 input still yields a proper orthogonal U up to the sign the caller fixes.
    double u[3][3];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) {
        double a = 0; for (int k = 0; k < 3; k++) a += A[i][k] * V[k][j];
        u[j][i] = a;
    }
    auto norm = [](double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    double n0 = norm(u[0]);
    if (n0 > 0) for (int i = 0; i < 3; i++) u[0][i] /= n0; else { u[0][0] = 1; u[0][1] = u[0][2] = 0; }
    double d01 = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
    for (int i = 0; i < 3; i++) u[1][i] -= d01 * u[0][i];
    double n1 = norm(u[1]);
    if (n1 > 1e-300 && sig[1] > 1e-14 * sig[0]) for (int i = 0; i < 3; i++) u[1][i] /= n1;
    else {  // pick any unit vector orthogonal to u0
        int m = std::fabs(u[0][0]) < std::fabs(u[0][1]) ? (std::fabs(u[0][0]) < std::fabs(u[0][2]) ? 0 : 2)
                                                       : (std::fabs(u[0][1]) < std::fabs(u[0][2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1;
        double d = u[0][m];
        for (int i = 0; i < 3; i++) u[1][i] = e[i] - d * u[0][i];
        n1 = norm(u[1]); for (int i = 0; i < 3; i++) u[1][i] /= n1;
    }
    double c[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2],
                   u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    // sign of the third column: align with A v_2 when that direction is resolvable
    double d2 = c[0] * u[2][0] + c[1] * u[2][1] + c[2] * u[2][2];
    double sgn = (sig[2] > 1e-14 * sig[0] && d2 < 0) ? -1.0 : 1.0;
    for (int i = 0; i < 3; i++) u[2][i] = sgn * c[i];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) U[i][j] = u[j][i];
}

double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// R = (U V^T)^T with the row-2 sign flip of V^T when det < 0: src/SingleThreadIcp.cpp:256-266.
void kabsch_R(const double H[3][3], double R[3][3]) {
    double U[3][3], V[3][3], s[3];
    svd3(H, U, s, V);
    auto make = [&](double sgn) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            // R[i][j] = (U Vt)^T[i][j] = sum_k U[j][k] * Vt[k][i],  Vt[k][i] = V[i][k], row k=2 scaled by sgn
            R[i][j] = U[j][0] * V[i][0] + U[j][1] * V[i][1] + sgn * U[j][2] * V[i][2];
        }
    };
    make(1.0);
    if (det3(R) < 0) make(-1.0);
}

// Row r of (M * [p;1]) accumulated k = 0..3 from a zero accumulator, multiply then add: the order
// Eigen's general matrix product uses for the 4x4 * 4xN products at src/SingleThreadIcp.cpp:277,302.
inline void apply4(const double M[16], const double* p, double* out) {
    for (int r = 0; r < 3; r++) {
        double acc = 0.0;
        acc += M[r * 4 + 0] * p[0];
        acc += M[r * 4 + 1] * p[1];
        acc += M[r * 4 + 2] * p[2];
        acc += M[r * 4 + 3] * 1.0;
        out[r] = acc;
    }
}

void matmul4(const double A[16], const double B[16], double C[16]) {
    double t[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double a = 0; for (int k = 0; k < 4; k++) a += A[i * 4 + k] * B[k * 4 + j];
        t[i * 4 + j] = a;
    }
    std::memcpy(C, t, sizeof t);
}

// ---------------------------------------------------------------------------------------------
// LDT: voxelisation box, splat, three-pass linear-time squared EDT, lookup.
// ---------------------------------------------------------------------------------------------
struct LdtBox { float mn[3]; float cellLen; };

// src/testLDT.cpp:50-69 with its arithmetic types: float bbox, max span * 1.1 in double, float
// mid-point ((min+max)/2 with float division, src/common.hpp:105-121), expansion factor 2.
LdtBox ldt_box(const float* pts, int64_t n, uint32_t div) {
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (int64_t i = 1; i < n; i++) for (int c = 0; c < 3; c++) {
        mn[c] = std::min(mn[c], pts[3 * i + c]); mx[c] = std::max(mx[c], pts[3 * i + c]);
    }
    float diag[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    float m01 = diag[0] < diag[1] ? diag[1] : diag[0];
    float m = m01 < diag[2] ? diag[2] : m01;
    double fullLen = (double)m * 1.1;
    double halfLen = fullLen / 2;
    LdtBox b;
    for (int c = 0; c < 3; c++) {
        float mid = (mn[c] + mx[c]) / (float)2;
        b.mn[c] = (float)((double)mid - halfLen * (double)2.0f);
    }
    fullLen = 2 * halfLen * (double)2.0f;
    b.cellLen = (float)(fullLen / (double)(float)div);
    return b;
}

}  // namespace

extern "C" {

// ---- NN ---------------------------------------------------------------------------------------

// Brute-force single nearest neighbour with lowest-index tie-break; distances as in
// src/kdtree.cpp:131-133. q: n x 3 doubles (row-major); tgt: m x 3 floats widened to double as at
// src/kdtree.cpp:60-70.
void orc_nn_brute(const double* q, int64_t n, const float* tgt, int64_t m, int32_t* idx, double* d2,
                  int threads) {
    std::vector<double> t(3 * m);
    for (int64_t j = 0; j < 3 * m; j++) t[j] = (double)tgt[j];
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double best = INFINITY; int32_t bi = -1;
        for (int64_t j = 0; j < m; j++) {
            double d = sqdist(q + 3 * i, t.data() + 3 * j);
            if (d < best) { best = d; bi = (int32_t)j; }
        }
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
    }
}

// Brute-force k nearest neighbours, ascending by (squared distance, index); slots beyond m get (-1, inf).
void orc_knn_brute(const double* q, int64_t n, const float* tgt, int64_t m, int k, int32_t* idx, double* d2, int threads) {
    std::vector<double> t(3 * m);
    for (int64_t j = 0; j < 3 * m; j++) t[j] = (double)tgt[j];
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        std::vector<double> bd(k, INFINITY);
        std::vector<int32_t> bi(k, -1);
        for (int64_t j = 0; j < m; j++) {
            const double d = sqdist(q + 3 * i, t.data() + 3 * j);
            if (!(d < bd[k - 1])) continue;          // equal distance: the earlier (lower) index stays
            int s = k - 1;
            while (s > 0 && d < bd[s - 1]) { bd[s] = bd[s - 1]; bi[s] = bi[s - 1]; s--; }
            bd[s] = d; bi[s] = (int32_t)j;
        }
        for (int s = 0; s < k; s++) { idx[i * k + s] = bi[s]; d2[i * k + s] = bd[s]; }
    }
}

// KDTree::k_close (src/kdtree.cpp:124-150) over n double queries: squared distances of the k nearest,
// ascending (the reference's priority queue holds them with the largest on top). Indices are not returned:
// among equal distances the reference keeps whichever node its traversal meets first.
void orc_kdtree_knn_d2(void* h, const double* q, int64_t n, int k, double* d2, int threads) {
    const KdTree* t = (const KdTree*)h;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        std::priority_queue<double> Q;
        t->search_k(q + 3 * i, 1, k, Q);
        for (int s = k - 1; s >= 0; s--) {
            if ((int)Q.size() > s) { d2[i * k + s] = Q.top(); Q.pop(); }
            else d2[i * k + s] = INFINITY;
        }
    }
}

void* orc_kdtree_build(const float* tgt, int64_t m) {
    KdTree* t = new KdTree;
    t->n = m;
    t->work.resize(m);
    for (int64_t i = 0; i < m; i++) {
        for (int c = 0; c < 3; c++) t->work[i].f[c] = (double)tgt[3 * i + c];
        t->work[i].index = (int)i; t->work[i].dim = 0;
    }
    t->heap.resize(4 * m + 8); t->used.assign(4 * m + 8, 0);
    if (m > 0) t->build(1, 0, m - 1, 0);
    return t;
}

void orc_kdtree_free(void* h) { delete (KdTree*)h; }

// KDTree::get_closest over n double queries (src/SingleThreadIcp.cpp:194-209 is the calling loop).
void orc_kdtree_nn(void* h, const double* q, int64_t n, int32_t* idx, double* d2, int threads) {
    const KdTree* t = (const KdTree*)h;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        KdTree::Best b{0.0, -1, false};
        t->search(q + 3 * i, 1, b);
        if (idx) idx[i] = b.index;
        if (d2) d2[i] = b.d2;
    }
}

// KDTree::Distance: float query widened to double, sqrt of the minimum, returned as float
// (src/kdtree.cpp:91-105).
void orc_kdtree_distance(void* h, const float* q, int64_t n, float* out, int threads) {
    const KdTree* t = (const KdTree*)h;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double p[3] = {(double)q[3 * i], (double)q[3 * i + 1], (double)q[3 * i + 2]};
        KdTree::Best b{0.0, -1, false};
        t->search(p, 1, b);
        out[i] = (float)std::sqrt(b.d2);
    }
}

// ---- SVD / Kabsch ---------------------------------------------------------------------------------

void orc_svd3(const double* A9, double* U9, double* s3, double* V9) {
    double A[3][3], U[3][3], V[3][3];
    std::memcpy(A, A9, sizeof A);
    svd3(A, U, s3, V);
    std::memcpy(U9, U, sizeof U); std::memcpy(V9, V, sizeof V);
}

void orc_kabsch_R(const double* H9, double* R9) {
    double H[3][3], R[3][3];
    std::memcpy(H, H9, sizeof H);
    kabsch_R(H, R);
    std::memcpy(R9, R, sizeof R);
}

// ---- ICP ----------------------------------------------------------------------------------------

// SingleThreadIcp::registration (src/SingleThreadIcp.cpp:293-327) with its constructor state
// (:49-59), cal_covarance (:216-247, including the target-mean carry-over at :229-231) and
// best_fit_transform (:249-290). nn_mode 0 = the reference's kd-tree, 1 = brute force lowest index.
// Outputs: per_iter_T (iters x 16, the per-iteration T), fin (16, finMat), mse (iters, the logged
// mean squared residual after each update, :281-287), idx0 (optional, correspondences of iteration 0).
static int icp_impl(const float* src, int64_t nd, const float* tgt, int64_t nm, const double* guess, int iters,
                    int nn_mode, double* per_iter_T, double* fin, double* mse_out, int32_t* idx_last, int threads,
                    double* mean_tar_io) {
    if (nd <= 0 || nm <= 0) return -1;
    std::vector<double> tar(3 * nm), cur(3 * nd), srcd(3 * nd);
    for (int64_t i = 0; i < 3 * nm; i++) tar[i] = (double)tgt[i];
    for (int64_t i = 0; i < 3 * nd; i++) srcd[i] = (double)src[i];
    // mean_tar is a member of the reference object: it is set by the constructor (cal_mean, :53) and
    // then carried from one registration() call to the next (:229-231 never resets it).
    double mean_tar[3] = {0, 0, 0};
    if (mean_tar_io && mean_tar_io[3] != 0.0) { for (int c = 0; c < 3; c++) mean_tar[c] = mean_tar_io[c]; }
    else {
        for (int64_t i = 0; i < nm; i++) for (int c = 0; c < 3; c++) mean_tar[c] += tar[3 * i + c];
        for (int c = 0; c < 3; c++) mean_tar[c] /= (double)nm;
    }

    void* kd = nn_mode == 0 ? orc_kdtree_build(tgt, nm) : nullptr;
    double finMat[16]; std::memcpy(finMat, guess, sizeof finMat);
    for (int64_t i = 0; i < nd; i++) apply4(guess, &srcd[3 * i], &cur[3 * i]);

    std::vector<int32_t> idx(nd);
    for (int it = 0; it < iters; it++) {
        if (nn_mode == 0) orc_kdtree_nn(kd, cur.data(), nd, idx.data(), nullptr, threads);
        else orc_nn_brute(cur.data(), nd, tgt, nm, idx.data(), nullptr, threads);
        // cal_covarance
        double mean_p[3] = {0, 0, 0};
        for (int64_t i = 0; i < nd; i++) for (int c = 0; c < 3; c++) mean_p[c] += cur[3 * i + c];
        for (int c = 0; c < 3; c++) mean_p[c] /= (double)nd;
        for (int64_t i = 0; i < nd; i++) for (int c = 0; c < 3; c++) mean_tar[c] += tar[3 * (int64_t)idx[i] + c];
        for (int c = 0; c < 3; c++) mean_tar[c] /= (double)nd;
        double H[3][3] = {{0}};
        for (int64_t i = 0; i < nd; i++) {
            double a[3], b[3];
            for (int c = 0; c < 3; c++) { a[c] = cur[3 * i + c] - mean_p[c]; b[c] = tar[3 * (int64_t)idx[i] + c] - mean_tar[c]; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r][c] += a[r] * b[c];
        }
        const double denom = (double)(float)nd;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r][c] /= denom;
        // best_fit_transform
        double R[3][3]; kabsch_R(H, R);
        double T[16] = {0};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r][c];
            T[r * 4 + 3] = mean_tar[r] - (R[r][0] * mean_p[0] + R[r][1] * mean_p[1] + R[r][2] * mean_p[2]);
        }
        T[15] = 1.0;
        double mse = 0.0;
        for (int64_t i = 0; i < nd; i++) {
            double o[3]; apply4(T, &cur[3 * i], o);
            cur[3 * i] = o[0]; cur[3 * i + 1] = o[1]; cur[3 * i + 2] = o[2];
            mse += (double)(float)sqdist(o, &tar[3 * (int64_t)idx[i]]);  // Dist() returns float: src/common.hpp:59-63
        }
        mse /= (double)nd;
        if (mse_out) mse_out[it] = mse;
        if (per_iter_T) std::memcpy(per_iter_T + 16 * it, T, sizeof T);
        matmul4(T, finMat, finMat);
    }
    if (idx_last) std::memcpy(idx_last, idx.data(), sizeof(int32_t) * nd);
    std::memcpy(fin, finMat, sizeof finMat);
    if (mean_tar_io) { for (int c = 0; c < 3; c++) mean_tar_io[c] = mean_tar[c]; mean_tar_io[3] = 1.0; }
    if (kd) orc_kdtree_free(kd);
    return 0;
}

int orc_icp(const float* src, int64_t nd, const float* tgt, int64_t nm, const double* guess, int iters,
            int nn_mode, double* per_iter_T, double* fin, double* mse_out, int32_t* idx_last, int threads) {
    return icp_impl(src, nd, tgt, nm, guess, iters, nn_mode, per_iter_T, fin, mse_out, idx_last, threads, nullptr);
}

// ---- DT3D convention (src/jly_3ddt.cpp:731-856) -----------------------------------------------------------
// Box rule of DT3D::Build (:732-764): info8 = SIZE, scale, xMin, yMin, zMin, xMax, yMax, zMax.
void orc_dt3d_box(const double* xyz, int64_t n, double expandFactor, double* info8) {
    double xMin = xyz[0], xMax = xyz[0], yMin = xyz[1], yMax = xyz[1], zMin = xyz[2], zMax = xyz[2];
    for (int64_t i = 1; i < n; i++) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (xMin > x) xMin = x;
        if (xMax < x) xMax = x;
        if (yMin > y) yMin = y;
        if (yMax < y) yMax = y;
        if (zMin > z) zMin = z;
        if (zMax < z) zMax = z;
    }
    const double xCenter = (xMin + xMax) / 2, yCenter = (yMin + yMax) / 2, zCenter = (zMin + zMax) / 2;
    xMin = xCenter - expandFactor * (xMax - xCenter);
    xMax = xCenter + expandFactor * (xMax - xCenter);
    yMin = yCenter - expandFactor * (yMax - yCenter);
    yMax = yCenter + expandFactor * (yMax - yCenter);
    zMin = zCenter - expandFactor * (zMax - zCenter);
    zMax = zCenter + expandFactor * (zMax - zCenter);
    double max = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
    max = max > zMax - zMin ? max : zMax - zMin;
    info8[0] = 300; info8[1] = 300 / max;
    info8[2] = xCenter - max / 2; info8[3] = yCenter - max / 2; info8[4] = zCenter - max / 2;
    info8[5] = xCenter + max / 2; info8[6] = yCenter + max / 2; info8[7] = zCenter + max / 2;
}

// Occupancy with ROUND(x) = int(x + 0.5) (:30, :792-804), stored [x][y][z] like the LDT grids here; returns
// the number of points skipped because they fall outside.
int64_t orc_dt3d_splat(const double* xyz, int64_t n, uint32_t size, const double* min3, double scale, uint8_t* occ) {
    const int S = (int)size;
    std::memset(occ, 0, (size_t)S * S * S);
    int64_t dropped = 0;
    for (int64_t i = 0; i < n; i++) {
        const int x = (int)((xyz[3 * i] - min3[0]) * scale + 0.5), y = (int)((xyz[3 * i + 1] - min3[1]) * scale + 0.5),
                  z = (int)((xyz[3 * i + 2] - min3[2]) * scale + 0.5);
        if (x < 0 || x >= S || y < 0 || y >= S || z < 0 || z >= S) { dropped++; continue; }
        occ[((size_t)x * S + y) * S + z] = 1;
    }
    return dropped;
}

// DT3D::Distance (:821-856) over a grid of exact squared cell distances sq[x][y][z].
void orc_dt3d_lookup(const int32_t* sq, uint32_t size, const double* min3, double scale, const double* q, int64_t n, float* out) {
    const int S = (int)size;
    for (int64_t i = 0; i < n; i++) {
        int x = (int)((q[3 * i] - min3[0]) * scale + 0.5), y = (int)((q[3 * i + 1] - min3[1]) * scale + 0.5),
            z = (int)((q[3 * i + 2] - min3[2]) * scale + 0.5);
        float a = 0, b = 0, c = 0;
        const bool inside = x > -1 && x < S && y > -1 && y < S && z > -1 && z < S;
        if (x < 0) { a = x; x = 0; } else if (x >= S) { a = x - S + 1; x = S - 1; }
        if (y < 0) { b = y; y = 0; } else if (y >= S) { b = y - S + 1; y = S - 1; }
        if (z < 0) { c = z; z = 0; } else if (z >= S) { c = z - S + 1; z = S - 1; }
        const float g = (float)(std::sqrt((double)sq[((size_t)x * S + y) * S + z]) / scale);
        out[i] = inside ? g : (float)(std::sqrt((double)(a * a + b * b + c * c)) / scale + g);
    }
}

// ---- ICP3D (src/jly_icp3d.cpp:181-295) ---------------------------------------------------------------
// The ICP of the original Go-ICP code that the reference carries: exact NN, optional trimming to the
// num = (int)(n * (1 - trim_fraction)) pairs with the smallest squared distance, stop when the summed
// error improves by less than err_diff * num. Restated as written, including: the two means are zeroed
// once and keep accumulating over the iterations, and are always divided by n (:204-206, :244-252,
// :262-263); R_ = V diag(1, 1, det(V U^T)) U^T with the determinant expanded along the first row
// (:271-285). The reference sorts with qsort and a comparator that never reports equality (:157-160), so
// its order among equal distances is unspecified; this restatement uses a stable sort (lower data index
// first). model / data: doubles, row-major; R9 / t3 in-out; returns err_new, *iterations = passes made.
double orc_icp3d_run(const double* model, int64_t m, const double* data, int64_t n, double* R9, double* t3, int64_t max_iter,
                     double err_diff, int do_trim, double trim_fraction, int threads, int64_t* iterations) {
    KdTree kd;
    kd.n = m;
    kd.work.resize(m);
    for (int64_t i = 0; i < m; i++) {
        for (int c = 0; c < 3; c++) kd.work[i].f[c] = model[3 * i + c];
        kd.work[i].index = (int)i; kd.work[i].dim = 0;
    }
    kd.heap.resize(4 * m + 8); kd.used.assign(4 * m + 8, 0);
    if (m > 0) kd.build(1, 0, m - 1, 0);
    if (threads > 0) omp_set_num_threads(threads);
    const int64_t num = do_trim ? (int64_t)(int)(n * (1 - trim_fraction)) : n;
    struct Ref { double dis; int64_t id_data; int id_model; };
    std::vector<Ref> pts(n);
    std::vector<double> p_m(3 * num), p_d(3 * num);
    double mu_m[3] = {0, 0, 0}, mu_d[3] = {0, 0, 0};
    double R[3][3], t[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i][j] = R9[3 * i + j]; t[i] = t3[i]; }
    double err = -1, err_new = 0;
    int64_t iter;
    for (iter = 0; iter < max_iter; iter++) {
        err_new = 0;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            double q[3];
            for (int r = 0; r < 3; r++) q[r] = R[r][0] * data[3 * i] + R[r][1] * data[3 * i + 1] + R[r][2] * data[3 * i + 2] + t[r];
            KdTree::Best b{0.0, -1, false};
            kd.search(q, 1, b);
            pts[i] = Ref{b.d2, i, b.index};
        }
        if (do_trim) std::stable_sort(pts.begin(), pts.end(), [](const Ref& a, const Ref& b) { return a.dis < b.dis; });
        for (int64_t i = 0; i < num; i++) {
            for (int c = 0; c < 3; c++) { p_m[3 * i + c] = model[3 * (int64_t)pts[i].id_model + c]; mu_m[c] += p_m[3 * i + c]; }
            const double* d = data + 3 * pts[i].id_data;
            for (int r = 0; r < 3; r++) { p_d[3 * i + r] = R[r][0] * d[0] + R[r][1] * d[1] + R[r][2] * d[2] + t[r]; mu_d[r] += p_d[3 * i + r]; }
            err_new += pts[i].dis;
        }
        if (err > 0 && err - err_new < err_diff * num) { iter++; break; }
        err = err_new;
        for (int c = 0; c < 3; c++) { mu_m[c] = mu_m[c] / (double)n; mu_d[c] = mu_d[c] / (double)n; }
        double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // ~q_t * q_m: rows = data, columns = model
        for (int64_t i = 0; i < num; i++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) H[r][c] += (p_d[3 * i + r] - mu_d[r]) * (p_m[3 * i + c] - mu_m[c]);
        double U[3][3], W[3], V[3][3], R_[3][3];
        svd3(H, U, W, V);
        auto compose = [&](double d) {
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R_[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + d * V[i][2] * U[j][2];
        };
        compose(1.0);
        const double a = R_[0][0] * (R_[1][1] * R_[2][2] - R_[1][2] * R_[2][1]);
        const double b = -R_[0][1] * (R_[1][0] * R_[2][2] - R_[1][2] * R_[2][0]);
        const double c = R_[0][2] * (R_[1][0] * R_[2][1] - R_[1][1] * R_[2][0]);
        compose(a + b + c);
        double t_[3], Rn[3][3], tn[3];
        for (int r = 0; r < 3; r++) t_[r] = mu_m[r] - (R_[r][0] * mu_d[0] + R_[r][1] * mu_d[1] + R_[r][2] * mu_d[2]);
        for (int r = 0; r < 3; r++) {
            for (int cc = 0; cc < 3; cc++) Rn[r][cc] = R_[r][0] * R[0][cc] + R_[r][1] * R[1][cc] + R_[r][2] * R[2][cc];
            tn[r] = (R_[r][0] * t[0] + R_[r][1] * t[1] + R_[r][2] * t[2]) + t_[r];
        }
        for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) R[r][cc] = Rn[r][cc]; t[r] = tn[r]; }
    }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R9[3 * i + j] = R[i][j]; t3[i] = t[i]; }
    if (iterations) *iterations = iter;
    return err_new;
}

// ---- LDT ------------------------------------------------------------------------------------------

void orc_ldt_box(const float* pts, int64_t n, uint32_t div, float* min3, float* cellLen) {
    LdtBox b = ldt_box(pts, n, div);
    min3[0] = b.mn[0]; min3[1] = b.mn[1]; min3[2] = b.mn[2]; *cellLen = b.cellLen;
}

// Splat (src/testLDT.cpp:74-88): float subtract, float divide, truncate; occupancy = count != 0.
// Returns the number of points whose voxel falls outside [0,div)^3 (the reference would write out
// of bounds for those; they are dropped here).
int64_t orc_ldt_splat(const float* pts, int64_t n, uint32_t div, const float* min3, float cellLen, uint8_t* occ) {
    std::memset(occ, 0, (size_t)div * div * div);
    int64_t dropped = 0;
    for (int64_t i = 0; i < n; i++) {
        int ix = (int)((pts[3 * i] - min3[0]) / cellLen);
        int iy = (int)((pts[3 * i + 1] - min3[1]) / cellLen);
        int iz = (int)((pts[3 * i + 2] - min3[2]) / cellLen);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= (int)div || iy >= (int)div || iz >= (int)div) { dropped++; continue; }
        occ[((size_t)ix * div + iy) * div + iz] = 1;
    }
    return dropped;
}

// Three-pass linear-time squared EDT (src/testLDT.cpp:91-206) on float grids like the reference,
// with the scan-back test of the y pass comparing y (not the outer x) against t[q] — the x pass at
// :197-202 shows the intended form; as committed (:152) the y pass reads s[-1], which is undefined
// behaviour and cannot be restated. Output: the squared grid before the final sqrt, as int32.
void orc_edt_meijster(const uint8_t* occ, uint32_t div_, int32_t* sq, int threads) {
    const int div = (int)div_;
    const size_t n3 = (size_t)div * div * div;
    std::vector<float> g(n3), dt(n3);
    const int infinity = 3 * div;
    auto at = [div](int x, int y, int z) { return ((size_t)x * div + y) * div + z; };
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for
    for (int x = 0; x < div; x++) for (int y = 0; y < div; y++) {
        g[at(x, y, 0)] = occ[at(x, y, 0)] ? 0.f : (float)infinity;
        for (int z = 1; z < div; z++) g[at(x, y, z)] = occ[at(x, y, z)] ? 0.f : g[at(x, y, z - 1)] + 1;
        for (int z = div - 2; z >= 0; z--) if (g[at(x, y, z + 1)] < g[at(x, y, z)]) g[at(x, y, z)] = 1 + g[at(x, y, z + 1)];
    }
#pragma omp parallel for
    for (int x = 0; x < div; x++) {
        std::vector<int> s(div), t(div);
        for (int z = 0; z < div; z++) {
            auto G = [&](int u) { float v = g[at(x, u, z)]; return v * v; };
            auto f = [&](int y, int u) { return (float)((y - u) * (y - u)) + G(u); };
            int q = 0; s[0] = 0; t[0] = 0;
            for (int u = 1; u < div; u++) {
                while (q >= 0 && f(t[q], s[q]) > f(t[q], u)) q--;
                if (q < 0) { q = 0; s[0] = u; }
                else {
                    int i = s[q];
                    int w = 1 + (int)(((float)(u * u - i * i) + G(u) - G(i)) / (float)(2 * (u - i)));
                    if (w < div) { q++; s[q] = u; t[q] = w; }
                }
            }
            for (int y = div - 1; y >= 0; y--) {
                dt[at(x, y, z)] = f(y, s[q]);
                if (y == t[q]) q--;
            }
        }
    }
#pragma omp parallel for
    for (int y = 0; y < div; y++) {
        std::vector<int> s(div), t(div);
        for (int z = 0; z < div; z++) {
            auto G = [&](int u) { return dt[at(u, y, z)]; };
            auto f = [&](int x, int u) { return (float)((x - u) * (x - u)) + G(u); };
            int q = 0; s[0] = 0; t[0] = 0;
            for (int u = 1; u < div; u++) {
                while (q >= 0 && f(t[q], s[q]) > f(t[q], u)) q--;
                if (q < 0) { q = 0; s[0] = u; }
                else {
                    int i = s[q];
                    int w = 1 + (int)(((float)(u * u - i * i) + G(u) - G(i)) / (float)(2 * (u - i)));
                    if (w < div) { q++; s[q] = u; t[q] = w; }
                }
            }
            for (int x = div - 1; x >= 0; x--) {
                sq[at(x, y, z)] = (int32_t)f(x, s[q]);
                if (x == t[q]) q--;
            }
        }
    }
}

// Definition-level check: min over occupied voxels of the squared index distance. O(V * occupied).
void orc_edt_brute(const uint8_t* occ, uint32_t div_, int32_t* sq, int threads) {
    const int div = (int)div_;
    std::vector<int> ox, oy, oz;
    for (int x = 0; x < div; x++) for (int y = 0; y < div; y++) for (int z = 0; z < div; z++)
        if (occ[((size_t)x * div + y) * div + z]) { ox.push_back(x); oy.push_back(y); oz.push_back(z); }
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for
    for (int x = 0; x < div; x++) for (int y = 0; y < div; y++) for (int z = 0; z < div; z++) {
        int64_t best = INT32_MAX;
        for (size_t k = 0; k < ox.size(); k++) {
            int64_t d = (int64_t)(x - ox[k]) * (x - ox[k]) + (int64_t)(y - oy[k]) * (y - oy[k]) + (int64_t)(z - oz[k]) * (z - oz[k]);
            if (d < best) best = d;
        }
        sq[((size_t)x * div + y) * div + z] = (int32_t)best;
    }
}

// Brute force at sampled voxels only (for the 300^3 grid): vox = ns x 3 int32 indices.
void orc_edt_brute_samples(const uint8_t* occ, uint32_t div_, const int32_t* vox, int64_t ns, int32_t* out) {
    const int div = (int)div_;
    std::vector<int> ox, oy, oz;
    for (int x = 0; x < div; x++) for (int y = 0; y < div; y++) for (int z = 0; z < div; z++)
        if (occ[((size_t)x * div + y) * div + z]) { ox.push_back(x); oy.push_back(y); oz.push_back(z); }
#pragma omp parallel for
    for (int64_t i = 0; i < ns; i++) {
        int x = vox[3 * i], y = vox[3 * i + 1], z = vox[3 * i + 2];
        int64_t best = INT32_MAX;
        for (size_t k = 0; k < ox.size(); k++) {
            int64_t d = (int64_t)(x - ox[k]) * (x - ox[k]) + (int64_t)(y - oy[k]) * (y - oy[k]) + (int64_t)(z - oz[k]) * (z - oz[k]);
            if (d < best) best = d;
        }
        out[i] = (int32_t)best;
    }
}

// LDT::Distance (src/testLDT.cpp:235-290) on the squared grid: g = sqrtf(sq) as the x pass stores it
// (:200). A coordinate equal to the upper clamp indexes div in the reference (out of range); it is
// clamped to div-1 here.
static inline float ldt_lookup(const int32_t* sq, int div, const float* mn, float cellLen, float x, float y, float z) {
    double dx = 0, dy = 0, dz = 0;
    float mx = mn[0] + cellLen * div, my = mn[1] + cellLen * div, mz = mn[2] + cellLen * div;
    if (x < mn[0]) { dx = mn[0] - x; x = mn[0]; }
    if (x > mx) { dx = x - mx; x = mx; }
    if (y < mn[1]) { dy = mn[1] - y; y = mn[1]; }
    if (y > my) { dy = y - my; y = my; }
    if (z < mn[2]) { dz = mn[2] - z; z = mn[2]; }
    if (z > mz) { dz = z - mz; z = mz; }
    int ix = (int)((x - mn[0]) / cellLen), iy = (int)((y - mn[1]) / cellLen), iz = (int)((z - mn[2]) / cellLen);
    ix = std::min(std::max(ix, 0), div - 1); iy = std::min(std::max(iy, 0), div - 1); iz = std::min(std::max(iz, 0), div - 1);
    float g = std::sqrt((float)sq[((size_t)ix * div + iy) * div + iz]);
    return (float)((double)(g * cellLen) + std::sqrt(dx * dx + dy * dy + dz * dz));
}

void orc_ldt_lookup(const int32_t* sq, uint32_t div, const float* min3, float cellLen, const float* q, int64_t n,
                    float* out, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++)
        out[i] = ldt_lookup(sq, (int)div, min3, cellLen, q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}

// ---- Go-ICP ---------------------------------------------------------------------------------------

#define ORC_PI 3.1415926536      // src/GoIcp.cpp:17 (truncated literal, kept)
#define ORC_SQRT3 1.732050808    // src/GoIcp.cpp:18
#define ORC_MAXROTLEVEL 20       // src/GoIcp.cpp:20

// GoIcp::Init's rotation-uncertainty table (src/GoIcp.cpp:462-480): out is 20 x n floats.
void orc_goicp_maxrotdist(const float* data, int64_t n, float* out) {
    // InitRot.w is a float member of RotNode (src/GoIcp.cpp:26), so the 2*PI stored at :88 is rounded to float
    // before :472 widens it again
    const double initRotW = (double)(float)(2 * ORC_PI);
    for (int l = 0; l < ORC_MAXROTLEVEL; l++) {
        double sigma = initRotW / std::pow(2.0, l) / 2.0;
        double tmp = sigma * ORC_SQRT3;
        if (tmp > ORC_PI) tmp = ORC_PI;
        for (int64_t j = 0; j < n; j++) {
            // SQ() of a float member is a float product and std::sqrt picks the float overload
            // (src/GoIcp.cpp:464-466); the result is then widened into the double table.
            float x = data[3 * j], y = data[3 * j + 1], z = data[3 * j + 2];
            double nrm = (double)std::sqrt(x * x + y * y + z * z);
            out[(size_t)l * n + j] = (float)(2 * std::sin(tmp / 2) * nrm);
        }
    }
}

// Rodrigues matrix of an angle-axis cube centre in float, as src/GoIcp.cpp:243-261 (the float
// overloads of sqrt/cos/sin are the ones selected there). R9 row-major. Returns 0 when t == 0
// (identity is used, see :274-279).
int orc_goicp_rotmat(float v1, float v2, float v3, float* R9) {
    float t = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (!(t > 0)) { for (int i = 0; i < 9; i++) R9[i] = (i % 4 == 0); return 0; }
    v1 /= t; v2 /= t; v3 /= t;
    float ct = std::cos(t), ct2 = 1 - ct, st = std::sin(t);
    float tmp121 = v1 * v2 * ct2, tmp122 = v3 * st;
    float tmp131 = v1 * v3 * ct2, tmp132 = v2 * st;
    float tmp231 = v2 * v3 * ct2, tmp232 = v1 * st;
    R9[0] = ct + v1 * v1 * ct2; R9[1] = tmp121 - tmp122; R9[2] = tmp131 + tmp132;
    R9[3] = tmp121 + tmp122; R9[4] = ct + v2 * v2 * ct2; R9[5] = tmp231 - tmp232;
    R9[6] = tmp131 - tmp132; R9[7] = tmp231 + tmp232; R9[8] = ct + v3 * v3 * ct2;
    return 1;
}

// Rotate the data cloud in float: x' = (R11 x + R12 y) + R13 z (src/GoIcp.cpp:266-272).
void orc_goicp_rotate(const float* R9, const float* data, int64_t n, float* out) {
    for (int64_t i = 0; i < n; i++) {
        float x = data[3 * i], y = data[3 * i + 1], z = data[3 * i + 2];
        out[3 * i] = R9[0] * x + R9[1] * y + R9[2] * z;
        out[3 * i + 1] = R9[3] * x + R9[4] * y + R9[5] * z;
        out[3 * i + 2] = R9[6] * x + R9[7] * y + R9[8] * z;
    }
}

struct DistFn {
    // mode 0: kd-tree exact distance (what the committed GoIcp uses: "KDTree dt", src/GoIcp.cpp:76-77)
    // mode 1: LDT lookup (the commented-out "LDT dt" alternative on the same lines)
    int mode; void* kd; const int32_t* sq; int div; float mn[3]; float cellLen;
    float operator()(float x, float y, float z) const {
        if (mode == 0) {
            double p[3] = {(double)x, (double)y, (double)z};
            KdTree::Best b{0.0, -1, false};
            ((const KdTree*)kd)->search(p, 1, b);
            return (float)std::sqrt(b.d2);
        }
        return ldt_lookup(sq, div, mn, cellLen, x, y, z);
    }
};

// Bound evaluation for a list of translation cubes under one rotated cloud: the loop body of
// GoIcp::innerBnB, src/GoIcp.cpp:384-431. cubes: nc x 4 floats (x,y,z,w) of the CHILD cubes.
// minDis_out (optional) nc x n; ub/lb are float sequential sums in ascending point order.
void orc_goicp_bounds(int mode, void* kd, const int32_t* sq, uint32_t div, const float* min3, float cellLen,
                      const float* rotated, int64_t n, const float* maxRotDistL, const float* cubes, int64_t nc,
                      float* ub_out, float* lb_out, float* minDis_out, int threads) {
    DistFn D{mode, kd, sq, (int)div, {0, 0, 0}, cellLen};
    if (min3) { D.mn[0] = min3[0]; D.mn[1] = min3[1]; D.mn[2] = min3[2]; }
    if (threads > 0) omp_set_num_threads(threads);
    std::vector<float> md(n);
    for (int64_t c = 0; c < nc; c++) {
        float w = cubes[4 * c + 3];
        float maxTransDis = (float)(ORC_SQRT3 / 2.0 * w);
        float tx = cubes[4 * c] + w / 2, ty = cubes[4 * c + 1] + w / 2, tz = cubes[4 * c + 2] + w / 2;
#pragma omp parallel for
        for (int64_t i = 0; i < n; i++) {
            float d = D(rotated[3 * i] + tx, rotated[3 * i + 1] + ty, rotated[3 * i + 2] + tz);
            if (maxRotDistL) d -= maxRotDistL[i];
            if (d < 0) d = 0;
            md[i] = d;
        }
        float ub = 0, lb = 0;
        for (int64_t i = 0; i < n; i++) ub += md[i] * md[i];
        for (int64_t i = 0; i < n; i++) { float dis = md[i] - maxTransDis; if (dis > 0) lb += dis * dis; }
        ub_out[c] = ub; lb_out[c] = lb;
        if (minDis_out) std::memcpy(minDis_out + (size_t)c * n, md.data(), sizeof(float) * n);
    }
}

struct TransNode { float x, y, z, w; double lb, ub; };
struct RotNode { float a, b, c, w; double lb, ub; int l; };
struct TransCmp { bool operator()(const TransNode& a, const TransNode& b) const { return a.lb > b.lb; } };
struct RotCmp { bool operator()(const RotNode& a, const RotNode& b) const { return a.lb > b.lb; } };

struct GoIcpState {
    DistFn D; const float* data; int64_t nd; const float* model; int64_t nm;
    std::vector<float> rotated, minDis, maxRot; float optError; float sseThresh; int threads;
    int64_t inner_evals = 0;
    double mean_tar_state[4] = {0, 0, 0, 0};   // {mean, valid flag}: the icp member's mean_tar between runICP calls
};

// GoIcp::innerBnB (src/GoIcp.cpp:353-457).
static float inner_bnb(GoIcpState& S, const float* maxRotDistL, TransNode* transOut) {
    std::priority_queue<TransNode, std::vector<TransNode>, TransCmp> que;
    TransNode init{-0.5f, -0.5f, -0.5f, 1.0f, 0, 0};
    que.push(init);
    float optErrorT = S.optError;
    const int64_t n = S.nd;
    while (!que.empty()) {
        TransNode parent = que.top(); que.pop();
        TransNode node;
        node.w = (float)(parent.w / 2.0);
        float maxTransDis = (float)(ORC_SQRT3 / 2.0 * node.w);
        for (int j = 0; j < 8; j++) {
            node.x = parent.x + (j & 1) * node.w;
            node.y = parent.y + (j >> 1 & 1) * node.w;
            node.z = parent.z + (j >> 2 & 1) * node.w;
            float tx = node.x + node.w / 2, ty = node.y + node.w / 2, tz = node.z + node.w / 2;
            S.inner_evals++;
#pragma omp parallel for
            for (int64_t i = 0; i < n; i++) {
                float d = S.D(S.rotated[3 * i] + tx, S.rotated[3 * i + 1] + ty, S.rotated[3 * i + 2] + tz);
                if (maxRotDistL) d -= maxRotDistL[i];
                if (d < 0) d = 0;
                S.minDis[i] = d;
            }
            float ub = 0, lb = 0;
            for (int64_t i = 0; i < n; i++) ub += S.minDis[i] * S.minDis[i];
            for (int64_t i = 0; i < n; i++) { float dis = S.minDis[i] - maxTransDis; if (dis > 0) lb += dis * dis; }
            if (ub < optErrorT) { optErrorT = ub; if (transOut) *transOut = node; }
            if (lb >= optErrorT) continue;
            node.lb = lb; node.ub = ub;
            if (optErrorT - node.lb < S.sseThresh) return optErrorT;
            que.push(node);
        }
    }
    return optErrorT;
}

// GoIcp::runICP (src/GoIcp.cpp:106-139): ICP (10 iterations, the constructor default at
// src/SingleThreadIcp.cpp:49) from a guess, then the float sequential SSE through the distance function.
static float run_icp(GoIcpState& S, double* M, int icp_iters, int nn_mode) {
    double fin[16];
    icp_impl(S.data, S.nd, S.model, S.nm, M, icp_iters, nn_mode, nullptr, fin, nullptr, nullptr, S.threads, S.mean_tar_state);
    std::memcpy(M, fin, sizeof fin);
    std::vector<float> err(S.nd);
#pragma omp parallel for
    for (int64_t i = 0; i < S.nd; i++) {
        double p[3] = {(double)S.data[3 * i], (double)S.data[3 * i + 1], (double)S.data[3 * i + 2]}, o[3];
        apply4(fin, p, o);
        float d = S.D((float)o[0], (float)o[1], (float)o[2]);
        err[i] = d * d;
    }
    float sse = 0; for (int64_t i = 0; i < S.nd; i++) sse += err[i];
    return sse;
}

// GoIcp::Init + OuterBnB (src/GoIcp.cpp:459-489, :142-350). run_bnb = 0 stops where the committed
// code does (the "return 0.0f" at :189, i.e. initial error + one ICP); run_bnb = 1 continues into the
// rotation search that follows it. dist_mode as DistFn. Outputs: optMat (16 doubles), stats
// {optError, initial error, post-ICP error, rotation cubes popped, inner evaluations, final lb}.
int orc_goicp(const float* model, int64_t nm, const float* data, int64_t nd, double mse, int run_bnb, int dist_mode,
              const int32_t* sq, uint32_t div, const float* min3, float cellLen, int icp_iters, int nn_mode,
              int64_t max_rot_pops, double* optMat, double* stats, int threads) {
    GoIcpState S;
    S.data = data; S.nd = nd; S.model = model; S.nm = nm; S.threads = threads;
    if (threads > 0) omp_set_num_threads(threads);
    void* kd = dist_mode == 0 ? orc_kdtree_build(model, nm) : nullptr;
    S.D = DistFn{dist_mode, kd, sq, (int)div, {0, 0, 0}, cellLen};
    if (min3) { S.D.mn[0] = min3[0]; S.D.mn[1] = min3[1]; S.D.mn[2] = min3[2]; }
    S.rotated.resize(3 * nd); S.minDis.resize(nd); S.maxRot.resize((size_t)ORC_MAXROTLEVEL * nd);
    orc_goicp_maxrotdist(data, nd, S.maxRot.data());
    const float mseThresh = (float)mse;
    S.sseThresh = (float)nd * mseThresh;
    double M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(optMat, M, sizeof M);

    S.optError = 0;
#pragma omp parallel for
    for (int64_t i = 0; i < nd; i++) S.minDis[i] = S.D(data[3 * i], data[3 * i + 1], data[3 * i + 2]);
    for (int64_t i = 0; i < nd; i++) S.optError += S.minDis[i] * S.minDis[i];
    stats[1] = S.optError;
    float error = run_icp(S, M, icp_iters, nn_mode);
    stats[2] = error;
    if (error < S.optError) { S.optError = error; std::memcpy(optMat, M, sizeof M); }
    stats[3] = 0; stats[5] = 0;
    if (run_bnb) {
        std::priority_queue<RotNode, std::vector<RotNode>, RotCmp> que;
        RotNode initRot; initRot.a = initRot.b = initRot.c = (float)-ORC_PI; initRot.w = (float)(2 * ORC_PI); initRot.l = 0; initRot.lb = 0; initRot.ub = 0;
        que.push(initRot);
        int64_t pops = 0; bool done = false;
        while (!done) {
            if (que.empty()) break;
            RotNode parent = que.top(); que.pop();
            stats[5] = parent.lb;
            if (S.optError - parent.lb < S.sseThresh) break;
            if (max_rot_pops > 0 && pops >= max_rot_pops) break;
            pops++;
            RotNode node; node.l = parent.l + 1; node.w = parent.w / 2;
            for (int j = 0; j < 8; j++) {
                node.a = parent.a + (j & 1) * node.w;
                node.b = parent.b + (j >> 1 & 1) * node.w;
                node.c = parent.c + (j >> 2 & 1) * node.w;
                float v1 = node.a + node.w / 2, v2 = node.b + node.w / 2, v3 = node.c + node.w / 2;
                if (std::sqrt(v1 * v1 + v2 * v2 + v3 * v3) - ORC_SQRT3 * node.w / 2 > ORC_PI) continue;
                float R[9]; orc_goicp_rotmat(v1, v2, v3, R);
                orc_goicp_rotate(R, data, nd, S.rotated.data());
                TransNode nt{-0.5f, -0.5f, -0.5f, 1.0f, 0, 0};
                float ub = inner_bnb(S, nullptr, &nt);
                if (ub < S.optError) {
                    S.optError = ub;
                    double O[16] = {R[0], R[1], R[2], nt.x + nt.w / 2, R[3], R[4], R[5], nt.y + nt.w / 2,
                                    R[6], R[7], R[8], nt.z + nt.w / 2, 0, 0, 0, 1};
                    std::memcpy(optMat, O, sizeof O);
                    double Mi[16]; std::memcpy(Mi, O, sizeof O);
                    float e = run_icp(S, Mi, icp_iters, nn_mode);
                    if (e < S.optError) { S.optError = e; std::memcpy(optMat, Mi, sizeof Mi); }
                    std::priority_queue<RotNode, std::vector<RotNode>, RotCmp> keep;
                    while (!que.empty()) { RotNode nn = que.top(); que.pop(); if (nn.lb < S.optError) keep.push(nn); else break; }
                    que = keep;
                }
                int lvl = std::min(node.l, ORC_MAXROTLEVEL - 1);
                float lb = inner_bnb(S, S.maxRot.data() + (size_t)lvl * nd, nullptr);
                if (lb >= S.optError) continue;
                node.lb = lb; node.ub = ub;
                if (S.optError - parent.lb < S.sseThresh) { done = true; break; }
                que.push(node);
            }
        }
        stats[3] = (double)pops;
    }
    stats[0] = S.optError; stats[4] = (double)S.inner_evals;
    if (kd) orc_kdtree_free(kd);
    return 0;
}

// ---- the icp/ fork's K-neighbour loop -----------------------------------------------------------------------
// icp() of icp/src/icp.hpp:398-460 with k_nearest_neighbors (:310-359) and the K-neighbour best_fit_transform
// (:185-253), made a function of its input: exact K nearest by (squared distance, index) of the float-rounded query
// (:335-336), the fork's comparator (:66-71, float difference truncated to int) applied by a stable bottom-up merge
// sort, and the fork's arithmetic statement by statement. Parity unpinned: the fork itself qsorts structs that own
// vectors (undefined behaviour), so there is no run of it to compare with.
int orc_knn_icp(const float* A, int64_t na, const float* B, int64_t nb, int max_iteration, float tolerance, int K, int remain,
                double* trans16, float* distances, int* iter_out, int threads) {
    if (na <= 0 || nb <= 0 || nb > na || K < 1 || K > 8 || nb < K) return -1;
    const int64_t n = nb;
    std::vector<double> P(3 * n);
    for (int64_t i = 0; i < 3 * n; i++) P[i] = (double)A[i];
    std::vector<int32_t> nidx((size_t)n * K), order(n), scratch(n);
    std::vector<double> nd2((size_t)n * K), qf(3 * n);
    std::vector<float> mean(n);
    double Tall[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double prev = 0.0;
    int it = 0;
    for (; it < max_iteration; it++) {
        for (int64_t i = 0; i < 3 * n; i++) qf[i] = (double)(float)P[i];
        orc_knn_brute(qf.data(), n, B, nb, K, nidx.data(), nd2.data(), threads);
        for (int64_t i = 0; i < n; i++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) acc += (float)nd2[i * K + k];
            mean[i] = acc / K;
            order[i] = (int32_t)i;
        }
        // merge sort of the source indices, runs doubling; an element of the right run goes first only when the
        // comparator says "before" ((int)(difference) < 0, i.e. difference <= -1)
        for (int64_t run = 1; run < n; run <<= 1) {
            for (int64_t s0 = 0; s0 < n; s0 += 2 * run) {
                const int64_t m = std::min(s0 + run, n), e = std::min(s0 + 2 * run, n);
                int64_t l = s0, r = m, o = s0;
                while (l < m && r < e) {
                    if (mean[order[r]] - mean[order[l]] <= -1.0f) scratch[o++] = order[r++];
                    else scratch[o++] = order[l++];
                }
                while (l < m) scratch[o++] = order[l++];
                while (r < e) scratch[o++] = order[r++];
            }
            order.swap(scratch);
        }
        const int num = (int)(n * remain / 100);
        double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
        for (int i = 0; i < num; i++) {
            const int32_t a = order[i];
            ca[0] += P[3 * a]; ca[1] += P[3 * a + 1]; ca[2] += P[3 * a + 2];
            for (int k = 0; k < K; k++) {
                const float* b = B + 3 * (int64_t)nidx[(int64_t)a * K + k];
                cb[0] += (double)b[0]; cb[1] += (double)b[1]; cb[2] += (double)b[2];
            }
            cb[0] /= K; cb[1] /= K; cb[2] /= K;
        }
        for (int c = 0; c < 3; c++) { ca[c] /= num; cb[c] /= num; }
        double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int64_t i = 0; i < n; i++) {
            double x[3], y[3];
            if (i < num) {
                const int32_t a = order[i];
                const float* b = B + 3 * (int64_t)nidx[(int64_t)a * K + (K - 1)];
                for (int c = 0; c < 3; c++) { x[c] = P[3 * a + c] - ca[c]; y[c] = ((double)b[c] - cb[c]) / K; }
            } else {
                for (int c = 0; c < 3; c++) x[c] = y[c] = P[3 * i + c];
            }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r][c] += x[r] * y[c];
        }
        double U[3][3], V[3][3], sg[3], R[3][3];
        svd3(H, U, sg, V);
        for (int pass = 0; pass < 2; pass++) {
            const double sgn = pass ? -1.0 : 1.0;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r][c] = U[r][0] * V[c][0] + U[r][1] * V[c][1] + sgn * U[r][2] * V[c][2];
            if (!(det3(R) < 0)) break;
        }
        double T[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 1}};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T[r][c] = R[r][c];
            T[r][3] = cb[r] - (R[r][0] * ca[0] + R[r][1] * ca[1] + R[r][2] * ca[2]);
        }
        double nxt[4][4];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
            double a = 0.0;
            for (int k = 0; k < 4; k++) a += T[r][k] * Tall[k][c];
            nxt[r][c] = a;
        }
        std::memcpy(Tall, nxt, sizeof nxt);
        for (int64_t i = 0; i < n; i++) {
            double o[3];
            for (int r = 0; r < 3; r++) {
                double a = 0.0;
                a += T[r][0] * P[3 * i]; a += T[r][1] * P[3 * i + 1]; a += T[r][2] * P[3 * i + 2]; a += T[r][3] * 1.0;
                o[r] = a;
            }
            P[3 * i] = o[0]; P[3 * i + 1] = o[1]; P[3 * i + 2] = o[2];
        }
        double err = 0.0;
        for (int64_t i = 0; i < n; i++) err += mean[order[i]];
        err /= (double)n;
        if (std::fabs(prev - err) < tolerance) break;
        prev = err;
    }
    std::memcpy(trans16, Tall, sizeof Tall);
    if (distances) for (int64_t i = 0; i < n; i++) distances[i] = mean[order[i]];
    *iter_out = it + 1;
    return 0;
}

int orc_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
