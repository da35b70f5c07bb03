// Host side of the registration classes (see icpx_host.hpp). Everything per-point runs on the GPU
// through include/icpx.h; what stays here is what the reference also does serially on the host:
// 3x3 SVD, 4x4 algebra, priority queues and the branch-and-bound decisions.
#include "icpx_host.hpp"
#include "../icpx_solve.h"
#include "icpx_io.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <queue>
#include <unordered_map>

namespace icp {

#define PI 3.1415926536        // src/GoIcp.cpp:17 (truncated literal kept on purpose)
#define SQRT3 1.732050808      // src/GoIcp.cpp:18
#define MAXROTLEVEL ICPX_MAXROTLEVEL

// ---- small algebra ------------------------------------------------------------------------------------

// The dense 3 x 3 / 4 x 4 arithmetic lives in icpx_solve.h, shared with the device-side registration loop.
Matrix4d Matrix4d::operator*(const Matrix4d& o) const {
    Matrix4d r;
    icpx_solve::mul4(m, o.m, r.m);
    return r;
}

static double det3(const Matrix3d& M) { return icpx_solve::det3(M.m); }

void svd3(const Matrix3d& A, Matrix3d& U, double sig[3], Matrix3d& V) { icpx_solve::svd3(A.m, U.m, sig, V.m); }

Matrix3d kabsch_rotation(const Matrix3d& H) {
    Matrix3d R;
    icpx_solve::kabsch_rotation(H.m, R.m);
    return R;
}

// Row r of M * [p;1], multiply then add from a zero accumulator in k order — the order of the 4x4 by
// 4xN products at src/SingleThreadIcp.cpp:277, :302, :320.
static inline void apply4(const Matrix4d& M, const double p[3], double out[3]) {
    for (int r = 0; r < 3; r++) {
        double a = 0.0;
        a += M.m[4 * r + 0] * p[0];
        a += M.m[4 * r + 1] * p[1];
        a += M.m[4 * r + 2] * p[2];
        a += M.m[4 * r + 3] * 1.0;
        out[r] = a;
    }
}

// ---- Device ---------------------------------------------------------------------------------------------

Device::Device(int device) {
    int rc = icpx_create(device, &ctx_);
    if (rc != ICPX_OK) throw IcpxFailure(rc, icpx_last_error(nullptr));
}
Device::~Device() { icpx_destroy(ctx_); }
void Device::check(int rc) const {
    if (rc != ICPX_OK) throw IcpxFailure(rc, icpx_last_error(ctx_));
}

static void ensure_target_on(const DevicePtr& dev, const PointCloudTPtr& cloud) {
    if (!cloud || cloud->points.empty()) throw IcpxFailure(ICPX_EINVAL, "empty target cloud");
    // The same cloud object (still alive, so its address cannot have been given to another one), same size, same storage:
    // already on the GPU. Points changed in place are not noticed: call Device::forget_target() after that.
    if (!dev->target_ref.expired() && dev->target_ref.lock() == cloud && dev->target_n == cloud->points.size() &&
        dev->target_first == cloud->points.data())
        return;
    dev->check(icpx_set_target(dev->ctx(), &cloud->points[0].x, cloud->points.size(), sizeof(Point3f), ICPX_HOST));
    dev->target_ref = cloud; dev->target_n = cloud->points.size(); dev->target_first = cloud->points.data();
    dev->edt_div = 0;
    dev->edt_owner = nullptr;
}

static void ensure_edt_on(const DevicePtr& dev, const PointCloudTPtr& cloud, uint32_t div) {
    ensure_target_on(dev, cloud);
    if (dev->edt_div == div && dev->edt_owner == nullptr) return;
    dev->check(icpx_edt_build(dev->ctx(), div));
    dev->edt_div = div;
    dev->edt_owner = nullptr;
}

// ---- KDTree ---------------------------------------------------------------------------------------------

KDTree::KDTree(const PointCloudTPtr& cloudptr, int _dims, DevicePtr d) : dims(_dims), target(cloudptr), dev(d ? d : std::make_shared<Device>(0)) {
    if (_dims != 3) throw IcpxFailure(ICPX_EINVAL, "KDTree: only 3 dimensions are supported");
    ensure_target_on(dev, target);
}

void KDTree::get_closest(const double* q, size_t n, int32_t* idx, double* d2) {
    ensure_target_on(dev, target);
    dev->check(icpx_nn(dev->ctx(), q, n, ICPX_HOST, idx, d2));
}

void KDTree::k_close(const double* q, size_t n, int k, int32_t* idx, double* d2) {
    ensure_target_on(dev, target);
    dev->check(icpx_knn(dev->ctx(), q, n, ICPX_HOST, k, idx, d2));
}

int KDTree::get_closest(const Vector3d& x) {
    int32_t idx = -1;
    get_closest(x.v, 1, &idx, nullptr);
    return idx;
}

void KDTree::Distance(const float* q, size_t n, float* out) {
    ensure_target_on(dev, target);
    dev->check(icpx_nn_distance(dev->ctx(), q, n, ICPX_HOST, out));
}

float KDTree::Distance(float x0, float x1, float x2) {
    const float q[3] = {x0, x1, x2};
    float out = 0;
    Distance(q, 1, &out);
    return out;
}

// ---- the icp/ fork's loop -----------------------------------------------------------------------------------

namespace {
struct KNeighbor {          // icp/src/icp.hpp:50-55, with the K target indices inline
    int sourceIndex;
    int targetIndexes[8];
    float distanceMean;
};

// cmpKNeighbor (icp/src/icp.hpp:66-71) returns the float difference truncated to int; "a before b" when that is negative
inline bool fork_before(const KNeighbor& a, const KNeighbor& b) {
    const float d = a.distanceMean - b.distanceMean;
    return d <= -1.0f;      // (int)d < 0, without the undefined conversion of a huge difference
}

// Stable bottom-up merge sort under that comparator (runs of 1, 2, 4, ...; the left run wins unless the right element is
// strictly before it). The comparator is not a strict weak order, so the result is defined by this procedure, not by
// "sortedness"; the CPU restatement in oracle/ follows the same procedure.
void fork_order(std::vector<KNeighbor>& v) {
    const size_t n = v.size();
    std::vector<KNeighbor> tmp(n);
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            const size_t mid = std::min(lo + w, n), hi = std::min(lo + 2 * w, n);
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = fork_before(v[j], v[i]) ? v[j++] : v[i++];
            while (i < mid) tmp[k++] = v[i++];
            while (j < hi) tmp[k++] = v[j++];
        }
        v.swap(tmp);
    }
}
}  // namespace

ICP_OUT knn_icp(const PointCloudTPtr& A, const PointCloudTPtr& B, int max_iteration, float tolerance, int K, int remainPercentage, DevicePtr dev) {
    if (!A || !B || A->points.empty() || B->points.empty()) throw IcpxFailure(ICPX_EINVAL, "knn_icp: empty cloud");
    if (K < 1 || K > 8) throw IcpxFailure(ICPX_EINVAL, "knn_icp: K must be in [1, 8]");
    if (remainPercentage < 0 || remainPercentage > 100) throw IcpxFailure(ICPX_EINVAL, "knn_icp: remainPercentage must be in [0, 100]");
    const size_t na = A->points.size(), nb = B->points.size();
    if (nb > na) throw IcpxFailure(ICPX_EINVAL, "knn_icp: the target cloud must not be larger than the source cloud (the fork reads beyond its matrix then)");
    if (nb < (size_t)K) throw IcpxFailure(ICPX_EINVAL, "knn_icp: fewer target points than K");
    const size_t row = nb;                                       // min(A.rows(), B.rows()), icp.hpp:401
    KDTree kd(B, 3, dev);
    std::vector<double> src(3 * row), q(3 * row), d2(row * (size_t)K);
    std::vector<int32_t> idx(row * (size_t)K);
    for (size_t i = 0; i < row; i++) { src[3 * i] = A->points[i].x; src[3 * i + 1] = A->points[i].y; src[3 * i + 2] = A->points[i].z; }
    auto Bp = [&](int j, int c) { const Point3f& p = B->points[j]; return (double)(c == 0 ? p.x : c == 1 ? p.y : p.z); };
    ICP_OUT out;
    Matrix4d T_all = Matrix4d::Identity();
    std::vector<KNeighbor> nb_list(row);
    double prev_error = 0, mean_error = 0;
    int iter = 0;
    for (iter = 0; iter < max_iteration; iter++) {
        // k_nearest_neighbors (:310-359): the query goes through float (:335-336); distances are squared (nanoflann's L2)
        for (size_t i = 0; i < 3 * row; i++) q[i] = (double)(float)src[i];
        kd.k_close(q.data(), row, K, idx.data(), d2.data());
        for (size_t i = 0; i < row; i++) {
            KNeighbor& nbr = nb_list[i];
            nbr.sourceIndex = (int)i;
            float meanDis = 0.0f;
            for (int k = 0; k < K; k++) { nbr.targetIndexes[k] = idx[i * K + k]; meanDis += (float)d2[i * K + k]; }
            nbr.distanceMean = meanDis / K;
        }
        fork_order(nb_list);
        // best_fit_transform (:185-253)
        const int num = (int)((long long)row * remainPercentage / 100);
        double cA[3] = {0, 0, 0}, cB[3] = {0, 0, 0};
        for (int i = 0; i < num; i++) {
            const int a = nb_list[i].sourceIndex;
            for (int c = 0; c < 3; c++) cA[c] += src[3 * a + c];
            for (int k = 0; k < K; k++) for (int c = 0; c < 3; c++) cB[c] += Bp(nb_list[i].targetIndexes[k], c);
            for (int c = 0; c < 3; c++) cB[c] /= K;              // :205, inside the point loop
        }
        for (int c = 0; c < 3; c++) { cA[c] /= num; cB[c] /= num; }
        Matrix3d H;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H(r, c) = 0.0;
        for (size_t i = 0; i < row; i++) {
            double aa[3], bb[3];
            if ((int)i < num) {
                const int a = nb_list[i].sourceIndex, b = nb_list[i].targetIndexes[K - 1];   // the last neighbour overwrites the others (:216-220)
                for (int c = 0; c < 3; c++) { aa[c] = src[3 * a + c] - cA[c]; bb[c] = (Bp(b, c) - cB[c]) / K; }
            } else {
                for (int c = 0; c < 3; c++) aa[c] = bb[c] = src[3 * i + c];                 // AA = BB = A where the loop does not reach (:192-193)
            }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H(r, c) += aa[r] * bb[c];
        }
        Matrix3d U, V, R;
        double sv[3];
        svd3(H, U, sv, V);
        auto make = [&](double sgn) {
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = U(r, 0) * V(c, 0) + U(r, 1) * V(c, 1) + sgn * U(r, 2) * V(c, 2);
        };
        make(1.0);
        if (det3(R) < 0) make(-1.0);
        Matrix4d T = Matrix4d::Identity();
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T(r, c) = R(r, c);
            T(r, 3) = cB[r] - (R(r, 0) * cA[0] + R(r, 1) * cA[1] + R(r, 2) * cA[2]);
        }
        T_all = T * T_all;
        for (size_t i = 0; i < row; i++) { double p[3]; apply4(T, &src[3 * i], p); src[3 * i] = p[0]; src[3 * i + 1] = p[1]; src[3 * i + 2] = p[2]; }
        mean_error = 0.0;
        for (size_t i = 0; i < row; i++) mean_error += nb_list[i].distanceMean;
        mean_error /= (double)row;
        if (std::fabs(prev_error - mean_error) < tolerance) break;
        prev_error = mean_error;
    }
    out.trans = T_all;
    out.distances.resize(row);
    for (size_t i = 0; i < row; i++) out.distances[i] = nb_list[i].distanceMean;
    out.iter = iter + 1;                                          // :457, also when the loop ran out
    return out;
}

// ---- LDT --------------------------------------------------------------------------------------------------

LDT::LDT(const PointCloudTPtr& cloud, uint32_t _div, int _core, DevicePtr d) : div(_div), CORES(_core), dev(d ? d : std::make_shared<Device>(0)), cloud_(cloud) {
    ensure_edt_on(dev, cloud, div);
    float mn[3];
    dev->check(icpx_edt_info(dev->ctx(), nullptr, mn, &cellLen, &occupied, nullptr));
    _min = Point3f(mn[0], mn[1], mn[2]);
    _max = Point3f(mn[0] + cellLen * div, mn[1] + cellLen * div, mn[2] + cellLen * div);
}

void LDT::Distance(const float* q, size_t n, float* out) {
    ensure_edt_on(dev, cloud_, div);      // another object on the same Device may have changed the target or the grid since
    dev->check(icpx_edt_lookup(dev->ctx(), q, n, ICPX_HOST, out));
}

float LDT::Distance(float x, float y, float z) {
    const float q[3] = {x, y, z};
    float out = 0;
    Distance(q, 1, &out);
    return out;
}

const std::vector<int32_t>& LDT::squared() {
    if (sq_.empty()) {
        ensure_edt_on(dev, cloud_, div);
        sq_.resize((size_t)div * div * div);
        dev->check(icpx_edt_read(dev->ctx(), sq_.data(), ICPX_HOST));
    }
    return sq_;
}

const std::vector<float>& LDT::grid() {
    if (g_.empty()) {
        const std::vector<int32_t>& s = squared();
        g_.resize(s.size());
        for (size_t i = 0; i < s.size(); i++) g_[i] = std::sqrt((float)s[i]);   // src/testLDT.cpp:200
    }
    return g_;
}

// ---- DT3D --------------------------------------------------------------------------------------------------

DT3D::DT3D(DevicePtr d) : SIZE(300), scale(0), expandFactor(2.0), xMin(0), xMax(0), yMin(0), yMax(0), zMin(0), zMax(0),
                          dev(d ? d : std::make_shared<Device>(0)) {}   // the reference leaves expandFactor unset; upstream Go-ICP uses 2

void DT3D::Build(const double* x, const double* y, const double* z, int num) {
    if (!x || !y || !z || num < 1) throw IcpxFailure(ICPX_EINVAL, "DT3D::Build: empty cloud");
    // Per axis (:732-756): extent of the coordinates, centre, then the extent stretched about the centre by
    // expandFactor; afterwards (:757-764) every axis gets the longest stretched side, again about its centre.
    struct Span { double lo, hi, mid; };
    auto stretched = [&](const double* v) {
        Span a{v[0], v[0], 0.0};
        for (int i = 1; i < num; i++) { a.lo = std::min(a.lo, v[i]); a.hi = std::max(a.hi, v[i]); }
        a.mid = (a.lo + a.hi) / 2;
        const double reach = expandFactor * (a.hi - a.mid);
        a.lo = a.mid - reach; a.hi = a.mid + reach;
        return a;
    };
    Span ax[3] = {stretched(x), stretched(y), stretched(z)};
    double side = ax[0].hi - ax[0].lo;
    for (int c = 1; c < 3; c++) side = std::max(side, ax[c].hi - ax[c].lo);
    for (Span& a : ax) { a.lo = a.mid - side / 2; a.hi = a.mid + side / 2; }
    xMin = ax[0].lo; xMax = ax[0].hi; yMin = ax[1].lo; yMax = ax[1].hi; zMin = ax[2].lo; zMax = ax[2].hi;
    SIZE = 300;
    scale = SIZE / side;
    cloud_ = std::make_shared<PointCloudT>();
    cloud_->points.resize((size_t)num);
    for (int i = 0; i < num; i++) cloud_->points[i] = Point3f((float)x[i], (float)y[i], (float)z[i]);
    ensure_grid();
}

// The context's grid is this object's (rounded voxel rule, this box): built now if another object has used the Device since.
void DT3D::ensure_grid() {
    if (!cloud_) throw IcpxFailure(ICPX_ESTATE, "DT3D::Distance before Build");
    ensure_target_on(dev, cloud_);
    if (dev->edt_owner == this) return;
    const double mn[3] = {xMin, yMin, zMin};
    dev->check(icpx_edt_build_round(dev->ctx(), (uint32_t)SIZE, mn, scale));
    dev->edt_div = 0;   // the context's grid no longer follows LDT's box rule
    dev->edt_owner = this;
}

void DT3D::Distance(const double* xyz, size_t n, float* out) {
    ensure_grid();
    dev->check(icpx_edt_lookup_round(dev->ctx(), xyz, n, ICPX_HOST, out));
}

float DT3D::Distance(double x, double y, double z) {
    const double q[3] = {x, y, z};
    float out = 0;
    Distance(q, 1, &out);
    return out;
}

// ---- ICP3D -------------------------------------------------------------------------------------------------

ICP3D::ICP3D(DevicePtr d) : err_diff_def(0.000001), max_iter_def(10000), trim_fraction(0), do_trim(true), dev(d ? d : std::make_shared<Device>(0)) {}

static PointCloudTPtr cloud_from_doubles(const double* xyz, size_t n) {
    auto c = std::make_shared<PointCloudT>();
    c->points.resize(n);
    for (size_t i = 0; i < n; i++) c->points[i] = Point3f((float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]);
    return c;
}

void ICP3D::Build(const double* model, size_t n) {
    if (!model || n == 0) throw IcpxFailure(ICPX_EINVAL, "ICP3D::Build: empty model");
    model_ = cloud_from_doubles(model, n);
    ensure_target_on(dev, model_);
}

double ICP3D::Run(const double* data, size_t n, Matrix3d& R, Vector3d& t) { return Run(data, n, R, t, max_iter_def, err_diff_def); }
double ICP3D::Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, size_t max_iter) { return Run(data, n, R, t, max_iter, err_diff_def); }
double ICP3D::Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, double err_diff) { return Run(data, n, R, t, max_iter_def, err_diff); }

double ICP3D::Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, size_t max_iter, double err_diff) {
    if (!model_) throw IcpxFailure(ICPX_ESTATE, "ICP3D::Run before Build");
    if (!data || n == 0) throw IcpxFailure(ICPX_EINVAL, "ICP3D::Run: empty data");
    ensure_target_on(dev, model_);
    if (!data_ || data_key_.size() != 3 * n || std::memcmp(data_key_.data(), data, 3 * n * sizeof(double)) != 0) {
        data_ = cloud_from_doubles(data, n);
        data_key_.assign(data, data + 3 * n);
    }
    dev->check(icpx_set_source(dev->ctx(), &data_->points[0].x, n, sizeof(Point3f), ICPX_HOST));
    const size_t num = do_trim ? (size_t)(int)((double)n * (1 - trim_fraction)) : n;   // :189-196
    // The means are set to zero once and then accumulate over the iterations (mu += ... at :244-252,
    // mu = mu / n at :262-263, always by n): reproduced.
    double mu_m[3] = {0, 0, 0}, mu_d[3] = {0, 0, 0};
    double err = -1, err_new = 0;
    iterations = 0;
    for (size_t iter = 0; iter < max_iter; iter++) {
        iterations = iter + 1;
        double T[16] = {R(0, 0), R(0, 1), R(0, 2), t[0], R(1, 0), R(1, 1), R(1, 2), t[1], R(2, 0), R(2, 1), R(2, 2), t[2], 0, 0, 0, 1};
        double sums[ICPX_NSUMS];
        dev->check(icpx_icp_step_trimmed(dev->ctx(), T, num, sums));
        for (int c = 0; c < 3; c++) { mu_d[c] += sums[c]; mu_m[c] += sums[3 + c]; }
        err_new = sums[15];
        if (err > 0 && err - err_new < err_diff * (double)num) break;   // :257
        err = err_new;
        for (int c = 0; c < 3; c++) { mu_m[c] /= (double)n; mu_d[c] /= (double)n; }
        // H = sum (p - mu_d)(q - mu_m)^T over the kept pairs (:264-268), from the sums
        Matrix3d H;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                H(r, c) = sums[6 + 3 * r + c] - mu_d[r] * sums[3 + c] - sums[r] * mu_m[c] + (double)num * mu_d[r] * mu_m[c];
        Matrix3d U, V;
        double w[3];
        svd3(H, U, w, V);
        Matrix3d R_;
        auto make = [&](double d) {
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R_(i, j) = V(i, 0) * U(j, 0) + V(i, 1) * U(j, 1) + d * V(i, 2) * U(j, 2);
        };
        make(1.0);
        make(det3(R_));   // R_ = V diag(1, 1, det(V U^T)) U^T (:271-285)
        double t_[3], Rn[9], tn[3];
        for (int r = 0; r < 3; r++) t_[r] = mu_m[r] - (R_(r, 0) * mu_d[0] + R_(r, 1) * mu_d[1] + R_(r, 2) * mu_d[2]);
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) Rn[3 * r + c] = R_(r, 0) * R(0, c) + R_(r, 1) * R(1, c) + R_(r, 2) * R(2, c);
            tn[r] = (R_(r, 0) * t[0] + R_(r, 1) * t[1] + R_(r, 2) * t[2]) + t_[r];
        }
        for (int k = 0; k < 9; k++) R.m[k] = Rn[k];
        for (int r = 0; r < 3; r++) t[r] = tn[r];
    }
    return err_new;
}

// ---- SingleThreadIcp -----------------------------------------------------------------------------------------

SingleThreadIcp::SingleThreadIcp(PointCloudTPtr _s, PointCloudTPtr _t, int _iter, DevicePtr d)
    : source(_s), target(_t), max_iter(_iter), dev(d ? d : std::make_shared<Device>(0)) {
    Nm = (int)target->points.size();
    setTarget(target);
    setSource(source);
}

void SingleThreadIcp::ensure_target() { ensure_target_on(dev, target); }

void SingleThreadIcp::setTarget(PointCloudTPtr _target) {
    target = _target;
    Nm = (int)target->points.size();
    ensure_target();
    // cal_mean(tar, mean_tar): sequential double sum of the widened floats (src/common.hpp:65-71)
    double s[3] = {0, 0, 0};
    for (const Point3f& p : target->points) { s[0] += (double)p.x; s[1] += (double)p.y; s[2] += (double)p.z; }
    for (int c = 0; c < 3; c++) mean_tar[c] = s[c] / (double)Nm;
}

void SingleThreadIcp::setSource(const PointCloudTPtr _source) {
    source = _source;
    Nd = (int)source->points.size();
    if (Nd == 0) throw IcpxFailure(ICPX_EINVAL, "empty source cloud");
    dev->check(icpx_set_source(dev->ctx(), &source->points[0].x, (size_t)Nd, sizeof(Point3f), ICPX_HOST));
    Nd_global = Nd;
    if (comm.world > 1 && comm.allreduce_sum_host) {
        double n = (double)Nd;
        comm.allreduce_sum_host(&n, 1, comm.user);
        Nd_global = (long long)n;
    }
    if (keep_correspondences) Neighbour.pointsPARA.resize(Nd);
}

void SingleThreadIcp::kdtree_search() {
    ensure_target();
    if (comm.world > 1) {
        if (!comm.allreduce_sum_host) throw IcpxFailure(ICPX_EINVAL, "Collective without allreduce_sum_host");
        dev->check(icpx_icp_step(dev->ctx(), finMat.m, sums_, keep_correspondences ? Neighbour.pointsPARA.data() : nullptr, ICPX_HOST));
        comm.allreduce_sum_host(sums_, ICPX_NSUMS, comm.user);
    } else {
        if (keep_correspondences) Neighbour.pointsPARA.resize(Nd);
        dev->check(icpx_icp_step(dev->ctx(), finMat.m, sums_, keep_correspondences ? Neighbour.pointsPARA.data() : nullptr, ICPX_HOST));
    }
}

void SingleThreadIcp::cal_covarance() {
    const double N = (double)Nd_global;
    // means, the reference's carry-over of the previous target mean, "/= (float)Nd": icpx_solve::covariance
    icpx_solve::covariance(sums_, N, (double)(float)Nd_global, mean_tar.v, mean_tmpdata.v, Neighbour.covariance.m);
    nn_mse_trace.push_back(sums_[15] / N);
}

Matrix4d& SingleThreadIcp::best_fit_transform() {
    double mse_after = 0;
    icpx_solve::best_fit(Neighbour.covariance.m, mean_tar.v, mean_tmpdata.v, sums_, (double)Nd_global, T.m, &mse_after);
    T_trace.push_back(T);
    mse_trace.push_back(mse_after);      // "each transform mse" (src/SingleThreadIcp.cpp:281-288)
    return T;
}

ICP_res SingleThreadIcp::registration(Matrix4d& guess, const PointCloudTPtr& pData) {
    if (pData.get() != source.get() || Nd != (int)pData->points.size()) setSource(pData);
    finMat = guess;
    mse_trace.clear();
    nn_mse_trace.clear();
    T_trace.clear();
    if (device_loop && comm.world <= 1 && !keep_correspondences && max_iter > 0) {
        // the whole loop on the device (icpx_icp_run): the same arithmetic, no round trip per iteration
        ensure_target();
        std::vector<double> trace(18 * (size_t)max_iter);
        dev->check(icpx_icp_run(dev->ctx(), guess.m, max_iter, mean_tar.v, finMat.m, trace.data()));
        for (int iter = 0; iter < max_iter; iter++) {
            const double* t = &trace[18 * (size_t)iter];
            std::memcpy(T.m, t, sizeof T.m);
            T_trace.push_back(T);
            mse_trace.push_back(t[16]);
            nn_mse_trace.push_back(t[17]);
        }
    } else
    for (int iter = 0; iter < max_iter; iter++) {
        kdtree_search();
        cal_covarance();
        const Matrix4d curMat = best_fit_transform();
        finMat = curMat * finMat;
    }
    ICP_res ans;
    ans.resN3.data.resize((size_t)Nd * 3);
    for (int i = 0; i < Nd; i++) {
        const Point3f& s = source->points[i];
        const double p[3] = {(double)s.x, (double)s.y, (double)s.z};
        apply4(finMat, p, &ans.resN3.data[3 * (size_t)i]);
    }
    ans.transMat = finMat;
    return ans;
}

// ---- GoIcp ----------------------------------------------------------------------------------------------------

GoIcp::GoIcp(PointCloudTPtr pModel_, PointCloudTPtr pData_, int _iter, double _mse, DevicePtr d, uint32_t dt_div)
    : pModel(pModel_), pData(pData_), mseThresh((float)_mse), icp(pData_, pModel_, 10, d ? d : std::make_shared<Device>(0)), ITER(_iter) {
    dev = icp.dev;
    InitRot.a = InitRot.b = InitRot.c = (float)-PI;
    InitRot.w = (float)(2 * PI);
    InitRot.l = 0;
    InitRot.lb = 0; InitRot.ub = 0;
    InitTrans.lb = 0; InitTrans.ub = 0;
    InitTrans.x = -0.5; InitTrans.y = -0.5; InitTrans.z = -0.5;
    InitTrans.w = 1.0;
    Nd = (int)pData->points.size();
    Nm = (int)pModel->points.size();
    minDis.resize(Nd);
    ensure_edt_on(dev, pModel, dt_div);
    dev->check(icpx_goicp_set_data(dev->ctx(), &pData->points[0].x, (size_t)Nd, sizeof(Point3f), ICPX_HOST));
    optError = 0; sseThresh = 0;
    optNodeRot = InitRot; optNodeTrans = InitTrans;
    for (int i = 0; i < 9; i++) R_[i] = (i % 4 == 0);
}

GoIcp::~GoIcp() { clear(); }

void GoIcp::Init() {
    clear();
    std::vector<float> table((size_t)MAXROTLEVEL * Nd);
    dev->check(icpx_goicp_read_maxrotdist(dev->ctx(), table.data()));
    maxRotDist = new float*[MAXROTLEVEL];
    for (int l = 0; l < MAXROTLEVEL; l++) {
        maxRotDist[l] = new float[Nd];
        std::memcpy(maxRotDist[l], table.data() + (size_t)l * Nd, sizeof(float) * Nd);
    }
    optNodeRot = InitRot;
    optNodeTrans = InitTrans;
    optMat = Matrix4d::Identity();
    sseThresh = Nd * mseThresh;
}

void GoIcp::clear() {
    if (!maxRotDist) return;
    for (int i = 0; i < MAXROTLEVEL; i++) delete[] maxRotDist[i];
    delete[] maxRotDist;
    maxRotDist = nullptr;
}

// d_i = dt.Distance((float)(M x_i)) for every data point: the exact nearest-neighbour distance (KDTree::Distance, what
// the committed reference's `KDTree dt` returns, src/GoIcp.cpp:76-77) or the distance-transform lookup (LDT::Distance).
void GoIcp::residuals(const Matrix4d& M, float* dist) {
    if (!exact_residuals) { dev->check(icpx_goicp_residuals(dev->ctx(), M.m, dist)); return; }
    std::vector<float> q((size_t)Nd * 3);
    for (int i = 0; i < Nd; i++) {
        const Point3f& s = pData->points[i];
        const double p[3] = {(double)s.x, (double)s.y, (double)s.z};
        double o[3];
        apply4(M, p, o);
        q[3 * (size_t)i] = (float)o[0]; q[3 * (size_t)i + 1] = (float)o[1]; q[3 * (size_t)i + 2] = (float)o[2];
    }
    dev->check(icpx_nn_distance(dev->ctx(), q.data(), (size_t)Nd, ICPX_HOST, dist));
}

float GoIcp::runICP(Matrix4d& tmpMat) {
    ICP_res r = icp.registration(tmpMat, pData);
    tmpMat = r.transMat;
    std::vector<float> dist(Nd);
    residuals(tmpMat, dist.data());
    float SSE = 0.0f;
    for (int i = 0; i < Nd; i++) SSE += dist[i] * dist[i];   // error[i] = dis*dis, then a float sum (:127, :133-135)
    return SSE;
}

int GoIcp::cur_level_of(const float* maxDistL) const {
    if (!maxDistL) return -1;
    if (maxRotDist)
        for (int l = 0; l < MAXROTLEVEL; l++) if (maxRotDist[l] == maxDistL) return l;
    throw IcpxFailure(ICPX_EINVAL, "innerBnB: maxDistL must be NULL or one of maxRotDist[level]");
}

namespace {
struct Stopwatch {
    double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~Stopwatch() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
double g_t_build = 0, g_t_take = 0, g_t_icp = 0;
long long g_pops_hist[12] = {0}, g_longest_sum = 0, g_run_calls = 0, g_longest_max = 0;   // ICPX_GOICP_VERBOSE: cubes popped per search (log2 buckets), longest search of each run_searches call   // ICPX_GOICP_VERBOSE: host time putting batches together, digesting bounds, in runICP
}  // namespace

// Evaluates a batch of cubes; with several ranks each one evaluates a contiguous slice and the slices
// are combined by a sum all-reduce (every rank then holds every bound and takes identical decisions).
void GoIcp::eval_items(std::vector<icpx_bnb_item>& items, std::vector<float>& ub, std::vector<float>& lb) {
    const size_t n = items.size();
    ub.assign(n, 0.f); lb.assign(n, 0.f);
    if (n == 0) return;
    inner_evals += (long long)n;
    launches++;
    Stopwatch watch{eval_seconds};
    const bool shared = shared_launches();
    size_t lo = 0, hi = n;
    if (shared) { lo = n * comm.rank / comm.world; hi = n * (comm.rank + 1) / comm.world; }
    std::vector<double> both(2 * n, 0.0);
    if (shared && comm.allreduce_sum_dev && comm.dev_buffer && comm.dev_capacity >= 16 && !parity_sums) {
        // Device side: every rank evaluates its share of the batch into the same layout (zeros elsewhere), the sum over
        // the ranks runs on the GPUs on the context's stream, and one copy brings all bounds to the host.
        const size_t cap = comm.dev_capacity / 2;
        for (size_t c0 = 0; c0 < n; c0 += cap) {
            const size_t m = std::min(cap, n - c0);
            const size_t a = m * comm.rank / comm.world, b = m * (comm.rank + 1) / comm.world;
            dev->check(icpx_bnb_eval_async(dev->ctx(), items.data() + c0, m, a, b - a, comm.dev_buffer));
            comm.allreduce_sum_dev(comm.dev_buffer, (int)(2 * m), comm.user);
            collectives++;
            std::vector<double> part(2 * m);
            dev->check(icpx_download(dev->ctx(), comm.dev_buffer, part.data(), 2 * m * sizeof(double)));
            for (size_t k = 0; k < m; k++) { ub[c0 + k] = (float)part[k]; lb[c0 + k] = (float)part[m + k]; }
        }
        return;
    }
    if (hi > lo) {
        const size_t m = hi - lo;
        if (parity_sums) {
            std::vector<double> u(m), l(m);
            std::vector<float> md(m * (size_t)Nd);
            dev->check(icpx_bnb_eval(dev->ctx(), items.data() + lo, m, u.data(), l.data(), md.data()));
            for (size_t k = 0; k < m; k++) {
                const float* d = md.data() + k * (size_t)Nd;
                const float maxTransDis = (float)(SQRT3 / 2.0 * items[lo + k].w);
                float fu = 0, fl = 0;
                for (int i = 0; i < Nd; i++) fu += d[i] * d[i];
                for (int i = 0; i < Nd; i++) { const float dis = d[i] - maxTransDis; if (dis > 0) fl += dis * dis; }
                both[lo + k] = fu; both[n + lo + k] = fl;
            }
        } else {
            dev->check(icpx_bnb_eval(dev->ctx(), items.data() + lo, m, both.data() + lo, both.data() + n + lo, nullptr));
        }
    }
    if (shared) {
        if (!comm.allreduce_sum_host) throw IcpxFailure(ICPX_EINVAL, "Collective without allreduce_sum_host");
        comm.allreduce_sum_host(both.data(), (int)(2 * n), comm.user);
        collectives++;
    }
    for (size_t k = 0; k < n; k++) { ub[k] = (float)both[k]; lb[k] = (float)both[n + k]; }
}

// One inner (translation) search: the state of GoIcp::innerBnB between two bound evaluations.
namespace {
struct TransQueue : std::priority_queue<TransNode> {
    const std::vector<TransNode>& heap() const { return c; }   // the binary heap itself, for looking ahead without popping
};
struct CubeKey {
    uint32_t k[4];
    bool operator==(const CubeKey& o) const { return std::memcmp(k, o.k, sizeof k) == 0; }
};
struct CubeKeyHash {
    size_t operator()(const CubeKey& c) const {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (uint32_t v : c.k) h = (h ^ v) * 0x100000001b3ull;
        return (size_t)h;
    }
};
CubeKey key_of(const TransNode& n) {
    CubeKey c;
    std::memcpy(&c.k[0], &n.x, 4); std::memcpy(&c.k[1], &n.y, 4); std::memcpy(&c.k[2], &n.z, 4); std::memcpy(&c.k[3], &n.w, 4);
    return c;
}
struct ChildBounds { float ub[8], lb[8]; };
}  // namespace

struct GoIcp::Search {
    float R[9];
    int level;                 // -1: upper-bound form
    TransQueue que;
    float optErrorT;
    TransNode best;
    bool have_best = false, done = false;
    int pops = 0;              // cubes taken off the queue so far
    std::vector<TransNode> pending;   // parents whose 8 children are in the batch, in evaluation order
    std::unordered_map<CubeKey, ChildBounds, CubeKeyHash> ahead;   // bounds of the children of cubes not popped yet
};

// The eight children of a translation cube in the reference's order (src/GoIcp.cpp:378-390).
static TransNode child_of(const TransNode& parent, int j) {
    TransNode node;
    node.w = (float)(parent.w / 2.0);
    node.x = parent.x + (j & 1) * node.w;
    node.y = parent.y + (j >> 1 & 1) * node.w;
    node.z = parent.z + (j >> 2 & 1) * node.w;
    node.lb = node.ub = 0;
    return node;
}

// Sends one batch of parent cubes (each stands for its eight children) to slot `slot`. Alone on its GPU the context
// evaluates it while the caller goes on (icpx_bnb_submit); with several ranks, or with the reference's sequential float
// sums, the batch is evaluated before this returns and the bounds wait in slot_ub / slot_lb.
void GoIcp::submit_parents(int slot, const std::vector<icpx_bnb_item>& parents) {
    const size_t n = parents.size();
    inner_evals += (long long)(8 * n);
    launches++;
    Stopwatch watch{eval_seconds};
    slot_count[slot] = n;
    slot_async[slot] = false;
    const bool shared = shared_launches();
    const bool dev_collective = shared && comm.allreduce_sum_dev && comm.dev_buffer && comm.dev_capacity >= 64;
    if (parity_sums || (shared && !dev_collective)) {
        std::vector<icpx_bnb_item> items;
        items.reserve(8 * n);
        for (const icpx_bnb_item& p : parents) {
            TransNode t; t.x = p.x; t.y = p.y; t.z = p.z; t.w = p.w; t.lb = t.ub = 0;
            for (int j = 0; j < 8; j++) {
                const TransNode c = child_of(t, j);
                icpx_bnb_item it = p;
                it.x = c.x; it.y = c.y; it.z = c.z; it.w = c.w;
                items.push_back(it);
            }
        }
        inner_evals -= (long long)(8 * n); launches--;      // eval_items counts them
        eval_items(items, slot_ub[slot], slot_lb[slot]);
        return;
    }
    if (shared) {
        // every rank evaluates its share of the parents into the same layout (zeros elsewhere), the sum over the ranks
        // runs on the GPUs on the context's stream, and one copy brings all bounds to the host. Each slot has its half
        // of the caller's device buffer; a batch that fits it stays in flight (the copy is only enqueued here), a
        // larger one is done piece by piece before this returns.
        const size_t half = comm.dev_capacity / 2 / 16 * 16;
        double* buf = comm.dev_buffer + (size_t)slot * half;
        const size_t a = n * comm.rank / comm.world, b = n * (comm.rank + 1) / comm.world;
        if (16 * n <= half) {
            dev->check(icpx_bnb_submit(dev->ctx(), slot, parents.data(), n, a, b - a, buf));
            comm.allreduce_sum_dev(buf, (int)(16 * n), comm.user);
            collectives++;
            dev->check(icpx_bnb_fetch(dev->ctx(), slot, buf));
            slot_async[slot] = true;
            return;
        }
        slot_ub[slot].assign(8 * n, 0.f); slot_lb[slot].assign(8 * n, 0.f);
        const size_t cap = half / 16;
        std::vector<double> part;
        for (size_t c0 = 0; c0 < n; c0 += cap) {
            const size_t m = std::min(cap, n - c0);
            const size_t a2 = m * comm.rank / comm.world, b2 = m * (comm.rank + 1) / comm.world;
            dev->check(icpx_bnb_submit(dev->ctx(), slot, parents.data() + c0, m, a2, b2 - a2, buf));
            comm.allreduce_sum_dev(buf, (int)(16 * m), comm.user);
            collectives++;
            part.resize(16 * m);
            dev->check(icpx_download(dev->ctx(), buf, part.data(), 16 * m * sizeof(double)));
            dev->check(icpx_bnb_collect(dev->ctx(), slot, nullptr, nullptr));
            for (size_t k = 0; k < 8 * m; k++) { slot_ub[slot][8 * c0 + k] = (float)part[k]; slot_lb[slot][8 * c0 + k] = (float)part[8 * m + k]; }
        }
        return;
    }
    // one submission holds at most 131070 parent cubes (icpx_bnb_submit): a larger batch goes in pieces, each waited for
    const size_t cap = 131070;
    if (n > cap) {
        slot_ub[slot].assign(8 * n, 0.f); slot_lb[slot].assign(8 * n, 0.f);
        std::vector<double> part;
        for (size_t c0 = 0; c0 < n; c0 += cap) {
            const size_t m = std::min(cap, n - c0);
            dev->check(icpx_bnb_submit(dev->ctx(), slot, parents.data() + c0, m, 0, m, nullptr));
            part.resize(16 * m);
            dev->check(icpx_bnb_collect(dev->ctx(), slot, part.data(), part.data() + 8 * m));
            for (size_t k = 0; k < 8 * m; k++) { slot_ub[slot][8 * c0 + k] = (float)part[k]; slot_lb[slot][8 * c0 + k] = (float)part[8 * m + k]; }
        }
        return;
    }
    dev->check(icpx_bnb_submit(dev->ctx(), slot, parents.data(), n, 0, n, nullptr));
    slot_async[slot] = true;
}

// In-place sum of host doubles over the ranks: the caller's host collective, or its device collective through the
// device buffer it provided (upload, sum on the context's stream, download), in pieces if the buffer is smaller.
void GoIcp::sum_over_ranks(std::vector<double>& v) {
    if (comm.world <= 1 || v.empty()) return;
    collectives++;
    if (comm.allreduce_sum_host) { comm.allreduce_sum_host(v.data(), (int)v.size(), comm.user); return; }
    if (!(comm.allreduce_sum_dev && comm.dev_buffer && comm.dev_capacity > 0)) throw IcpxFailure(ICPX_EINVAL, "Collective with neither a host nor a device all-reduce");
    for (size_t c0 = 0; c0 < v.size(); c0 += comm.dev_capacity) {
        const size_t m = std::min(comm.dev_capacity, v.size() - c0);
        dev->check(icpx_upload(dev->ctx(), comm.dev_buffer, v.data() + c0, m * sizeof(double)));
        comm.allreduce_sum_dev(comm.dev_buffer, (int)m, comm.user);
        dev->check(icpx_download(dev->ctx(), comm.dev_buffer, v.data() + c0, m * sizeof(double)));
    }
}

void GoIcp::collect_parents(int slot, std::vector<float>& ub, std::vector<float>& lb) {
    if (!slot_async[slot]) { ub.swap(slot_ub[slot]); lb.swap(slot_lb[slot]); return; }
    Stopwatch watch{eval_seconds};
    const size_t n = 8 * slot_count[slot];
    slot_d.resize(2 * n);
    dev->check(icpx_bnb_collect(dev->ctx(), slot, slot_d.data(), slot_d.data() + n));
    ub.resize(n); lb.resize(n);
    for (size_t k = 0; k < n; k++) { ub[k] = (float)slot_d[k]; lb[k] = (float)slot_d[n + k]; }
}

// The searches as one call of icpx_bnb_search_run. Searches the device could not finish (a queue beyond its capacity, too
// many steps) are reset to their start and handed back in `left`.
bool GoIcp::run_searches_device(const std::vector<Search*>& ss, std::vector<Search*>& left) {
    if (ss.size() > ICPX_SEARCH_MAX) {               // more than one call takes: in pieces
        for (size_t lo = 0; lo < ss.size(); lo += ICPX_SEARCH_MAX) {
            const std::vector<Search*> piece(ss.begin() + lo, ss.begin() + std::min(ss.size(), lo + (size_t)ICPX_SEARCH_MAX));
            if (!run_searches_device(piece, left)) return false;
        }
        return true;
    }
    std::vector<icpx_bnb_search> in(ss.size());
    std::vector<icpx_bnb_found> out(ss.size());
    for (size_t i = 0; i < ss.size(); i++) {
        const Search& s = *ss[i];
        if (s.que.size() != 1 || s.pops != 0 || s.done) return false;      // not at its start: the host loop carries on
        std::memcpy(in[i].R, s.R, sizeof s.R);
        in[i].level = s.level;
        in[i].opt_error = s.optErrorT;
        const TransNode& t = s.que.top();
        in[i].cube[0] = t.x; in[i].cube[1] = t.y; in[i].cube[2] = t.z; in[i].cube[3] = t.w;
    }
    long long steps = 0;
    {
        Stopwatch watch{eval_seconds};
        dev->check(icpx_bnb_search_run(dev->ctx(), in.data(), in.size(), sseThresh, out.data(), &steps));
    }
    search_steps += steps;
    launches += steps;
    for (size_t i = 0; i < ss.size(); i++) {
        Search& s = *ss[i];
        inner_evals += 8LL * out[i].cubes;
        if (out[i].status != 0) { left.push_back(&s); continue; }
        s.optErrorT = out[i].opt_error;
        s.have_best = out[i].have_best != 0;
        if (s.have_best) { s.best.x = out[i].best[0]; s.best.y = out[i].best[1]; s.best.z = out[i].best[2]; s.best.w = out[i].best[3]; s.best.lb = s.best.ub = 0; }
        s.pops = out[i].cubes;
        s.done = true;
        while (!s.que.empty()) s.que.pop();
    }
    return true;
}

void GoIcp::run_searches(const std::vector<Search*>& ss) {
    if (ss.empty()) return;
    if (device_search && frontier_width == 1 && !parity_sums && !shared_launches()) {
        std::vector<Search*> left;
        if (run_searches_device(ss, left)) {
            if (!left.empty()) run_searches_host(left);
            return;
        }
    }
    run_searches_host(ss);
}

void GoIcp::run_searches_host(const std::vector<Search*>& ss) {
    const int width = std::max(1, frontier_width);
    const int ahead_base = width == 1 ? std::max(0, lookahead) : 0;
    std::vector<size_t> cand;    // look-ahead: heap positions still to visit
    // Takes the children's bounds of one popped parent in the reference's order (src/GoIcp.cpp:403-452).
    auto take = [&](Search& s, const TransNode& parent, const float* u8, const float* l8) {
        s.pops++;
        for (int j = 0; j < 8 && !s.done; j++) {
            TransNode node = child_of(parent, j);
            const float u = u8[j], l = l8[j];
            if (u < s.optErrorT) { s.optErrorT = u; s.best = node; s.have_best = true; }
            if (l >= s.optErrorT) continue;
            node.lb = l; node.ub = u;
            if (s.optErrorT - node.lb < sseThresh) { s.done = true; continue; }
            s.que.push(node);
        }
    };
    // The searches are independent of each other, so they advance as two groups, each with a batch of its own: while
    // the GPU evaluates one group's batch (and, with several ranks, sums it over the ranks) the host digests the other's
    // bounds and puts its next batch together. With a host-side collective or sequential float sums a batch is evaluated
    // where it is submitted and one group does.
    const bool host_collective = shared_launches() && !(comm.allreduce_sum_dev && comm.dev_buffer && comm.dev_capacity >= 64);
    const int ngroups = (pipeline && !host_collective && !parity_sums && ss.size() >= 2) ? 2 : 1;
    struct Group { size_t lo, hi; std::vector<icpx_bnb_item> parents; bool in_flight = false; };
    Group groups[2];
    groups[0].lo = 0; groups[0].hi = ngroups == 2 ? ss.size() / 2 : ss.size();
    groups[1].lo = groups[0].hi; groups[1].hi = ss.size();
    std::vector<float> ub, lb;

    // The next batch of a group: for every search that still runs, the cube on top of its queue (after those whose
    // children's bounds are known already) and the cubes looked ahead.
    auto build = [&](Group& g) {
        Stopwatch w_{g_t_build};
        g.parents.clear();
        // When only a few searches of the group are still running, a step is a round trip for a handful of cubes: they
        // look further ahead then (up to 32 cubes each, about 64 parent cubes per batch), which costs little on so
        // small a batch and saves most of the remaining steps. Looking ahead changes no decision.
        int ahead = ahead_base;
        if (width == 1 && tail_lookahead) {
            size_t active = 0;
            for (size_t si = g.lo; si < g.hi; si++) active += !ss[si]->done;
            if (active > 0 && active * (size_t)(1 + ahead_base) < 64) ahead = std::max(ahead_base, (int)std::min<size_t>(32, 64 / active));
        }
        for (size_t si = g.lo; si < g.hi; si++) {
            Search& s = *ss[si];
            s.pending.clear();
            if (s.done) continue;
            // parents whose children were evaluated ahead of time need no launch
            while (ahead && !s.done && !s.que.empty()) {
                auto it = s.ahead.find(key_of(s.que.top()));
                if (it == s.ahead.end()) break;
                const TransNode parent = s.que.top(); s.que.pop();
                const ChildBounds cb = it->second;
                s.ahead.erase(it);
                take(s, parent, cb.ub, cb.lb);
            }
            if (s.done) continue;
            if (s.que.empty()) { s.done = true; continue; }
            for (int k = 0; k < width && !s.que.empty(); k++) {
                s.pending.push_back(s.que.top()); s.que.pop();
            }
            if (ahead) {
                // the `ahead` best cubes left in the heap, found from its root without disturbing it
                const std::vector<TransNode>& h = s.que.heap();
                cand.clear();
                if (!h.empty()) cand.push_back(0);
                for (int k = 0; k < ahead && !cand.empty(); k++) {
                    size_t bi = 0;
                    for (size_t c = 1; c < cand.size(); c++) if (h[cand[c]].lb < h[cand[bi]].lb) bi = c;
                    const size_t i = cand[bi];
                    cand[bi] = cand.back(); cand.pop_back();
                    if (2 * i + 1 < h.size()) cand.push_back(2 * i + 1);
                    if (2 * i + 2 < h.size()) cand.push_back(2 * i + 2);
                    if (!s.ahead.count(key_of(h[i]))) s.pending.push_back(h[i]);
                }
            }
            for (const TransNode& parent : s.pending) {
                icpx_bnb_item it;
                std::memcpy(it.R, s.R, sizeof it.R);
                it.x = parent.x; it.y = parent.y; it.z = parent.z; it.w = parent.w; it.level = s.level;
                g.parents.push_back(it);
            }
        }
    };
    auto digest = [&](Group& g) {
        Stopwatch w_{g_t_take};
        size_t pos = 0;
        for (size_t si = g.lo; si < g.hi; si++) {
            Search& s = *ss[si];
            const size_t popped = std::min<size_t>(s.pending.size(), (size_t)width);
            for (size_t k = 0; k < s.pending.size(); k++, pos += 8) {
                if (k < popped) { if (!s.done) take(s, s.pending[k], &ub[pos], &lb[pos]); }   // an earlier parent of this batch may have ended the search
                else {
                    ChildBounds cb;
                    std::memcpy(cb.ub, &ub[pos], sizeof cb.ub); std::memcpy(cb.lb, &lb[pos], sizeof cb.lb);
                    s.ahead.emplace(key_of(s.pending[k]), cb);
                }
            }
        }
    };
    for (int g = 0; g < ngroups; g++) {
        build(groups[g]);
        if (!groups[g].parents.empty()) { submit_parents(g, groups[g].parents); groups[g].in_flight = true; }
    }
    while (groups[0].in_flight || groups[1].in_flight)
        for (int g = 0; g < ngroups; g++) {
            if (!groups[g].in_flight) continue;
            collect_parents(g, ub, lb);
            groups[g].in_flight = false;
            digest(groups[g]);
            build(groups[g]);
            if (!groups[g].parents.empty()) { submit_parents(g, groups[g].parents); groups[g].in_flight = true; }
        }
    int longest = 0;
    for (const Search* s : ss) {
        int b = 0;
        while ((1 << b) < s->pops && b < 11) b++;
        g_pops_hist[b]++;
        longest = std::max(longest, s->pops);
    }
    g_longest_sum += longest; g_run_calls++; g_longest_max = std::max<long long>(g_longest_max, longest);
}

float GoIcp::innerBnB(float* maxDistL, TransNode* transOut) {
    Search s;
    std::memcpy(s.R, R_, sizeof R_);
    s.level = cur_level_of(maxDistL);
    s.que.push(InitTrans);
    s.optErrorT = optError;
    const int keep = frontier_width;
    frontier_width = 1;                 // the reference pops one parent at a time
    run_searches(std::vector<Search*>{&s});
    frontier_width = keep;
    if (transOut && s.have_best) *transOut = s.best;
    return s.optErrorT;
}

// Rodrigues matrix of the cube centre in float (src/GoIcp.cpp:243-261): R = cos t I + (1 - cos t) n n^T + sin t [n]x
// for the unit axis n = v / t, every entry formed as (n_i n_j)(1 - cos t) first and the cos / sin term added or
// subtracted last, which is the rounding order of the reference's temporaries. Returns false when t == 0.
static bool rotation_of(float v1, float v2, float v3, float R[9]) {
    float n[3] = {v1, v2, v3};
    const float t = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(t > 0)) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0); return false; }
    for (float& c : n) c /= t;
    const float cs = std::cos(t), vers = 1 - cs, sn = std::sin(t);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const float outer = n[std::min(i, j)] * n[std::max(i, j)] * vers;
            if (i == j) { R[3 * i + j] = cs + outer; continue; }
            const float cross = n[3 - i - j] * sn;                     // the axis component that is neither i nor j
            R[3 * i + j] = (j - i + 3) % 3 == 1 ? outer - cross : outer + cross;
        }
    return true;
}

float GoIcp::OuterBnB() {
    rot_pops = inner_evals = launches = collectives = search_steps = 0;
    eval_seconds = 0;

    // initial error: identity pose, sequential float sum (src/GoIcp.cpp:159-172)
    {
        Matrix4d I = Matrix4d::Identity();
        residuals(I, minDis.data());
        optError = 0;
        for (int i = 0; i < Nd; i++) optError += minDis[i] * minDis[i];
        initError = optError;
    }
    Matrix4d tmpMat = optMat;
    float error = runICP(tmpMat);
    icpError = error;
    if (error < optError) { optError = error; optMat = tmpMat; }
    final_lb = 0;
    if (!run_bnb) return 0.0f;   // the committed code returns here (src/GoIcp.cpp:189)

    // Work of one rotation parent: its child cubes inside the pi-ball with their rotations, and (batched mode) the two
    // inner searches of every child — upper-bound form, lower-bound form — started from the optError of that moment.
    struct Child { RotNode node; float R[9]; };
    struct Work { RotNode parent; std::vector<Child> kids; std::vector<Search> ss; float optAtStart; };
    auto same_cube = [](const RotNode& x, const RotNode& y) {
        return std::memcmp(&x.a, &y.a, sizeof(float)) == 0 && std::memcmp(&x.b, &y.b, sizeof(float)) == 0 &&
               std::memcmp(&x.c, &y.c, sizeof(float)) == 0 && std::memcmp(&x.w, &y.w, sizeof(float)) == 0 && x.l == y.l;
    };
    auto prepare = [&](const RotNode& parent, Work& w) {
        w.parent = parent;
        w.optAtStart = optError;
        RotNode nodeRot;
        nodeRot.l = parent.l + 1;
        nodeRot.w = parent.w / 2;
        nodeRot.lb = nodeRot.ub = 0;
        for (int j = 0; j < 8; j++) {
            nodeRot.a = parent.a + (j & 1) * nodeRot.w;
            nodeRot.b = parent.b + (j >> 1 & 1) * nodeRot.w;
            nodeRot.c = parent.c + (j >> 2 & 1) * nodeRot.w;
            const float v1 = nodeRot.a + nodeRot.w / 2, v2 = nodeRot.b + nodeRot.w / 2, v3 = nodeRot.c + nodeRot.w / 2;
            if (std::sqrt(v1 * v1 + v2 * v2 + v3 * v3) - SQRT3 * nodeRot.w / 2 > PI) continue;   // outside the pi-ball
            Child c;
            c.node = nodeRot;
            rotation_of(v1, v2, v3, c.R);
            w.kids.push_back(c);
        }
        if (!batched) return;
        w.ss.resize(2 * w.kids.size());
        for (size_t k = 0; k < w.kids.size(); k++)
            for (int form = 0; form < 2; form++) {
                Search& s = w.ss[2 * k + form];
                std::memcpy(s.R, w.kids[k].R, sizeof s.R);
                s.level = form == 0 ? -1 : std::min(w.kids[k].node.l, MAXROTLEVEL - 1);
                s.que.push(InitTrans);
                s.optErrorT = optError;
            }
    };
    struct RotQueue : std::priority_queue<RotNode> {
        const std::vector<RotNode>& heap() const { return c; }
    };
    RotQueue rq;
    std::vector<std::unique_ptr<Work>> ahead_of_time;   // searched already: valid while optError stays what it was
    spec_hits = spec_misses = 0;

    rq.push(InitRot);
    for (;;) {
        if (rq.empty()) break;
        const RotNode parent = rq.top(); rq.pop();
        final_lb = parent.lb;
        if (optError - parent.lb < sseThresh) break;
        if (max_rot_pops > 0 && rot_pops >= max_rot_pops) break;
        rot_pops++;

        std::unique_ptr<Work> cur;
        if (batched) {
            // what was searched ahead of time stands only if optError is still the value those searches started from
            for (size_t k = 0; k < ahead_of_time.size();) {
                if (std::memcmp(&ahead_of_time[k]->optAtStart, &optError, sizeof(float)) != 0) {
                    spec_misses++;
                    ahead_of_time.erase(ahead_of_time.begin() + k);
                } else k++;
            }
            for (size_t k = 0; k < ahead_of_time.size(); k++)
                if (same_cube(ahead_of_time[k]->parent, parent)) {
                    cur = std::move(ahead_of_time[k]);
                    ahead_of_time.erase(ahead_of_time.begin() + k);
                    spec_hits++;
                    break;
                }
            std::vector<Search*> run;
            std::vector<Work*> fresh;     // the cubes whose searches run in this step, in a fixed order (every rank forms the same list)
            if (!cur) {
                cur.reset(new Work);
                prepare(parent, *cur);
                for (Search& s : cur->ss) run.push_back(&s);
                fresh.push_back(cur.get());
            }
            // With them, in the same launches, the cubes most likely to be popped next: the best of the queue as it
            // stands. Only when a search has to run anyway: a cube found searched already costs no launch at all.
            if (speculate > 0 && !run.empty()) {
                const std::vector<RotNode>& h = rq.heap();
                std::vector<size_t> cand;
                if (!h.empty()) cand.push_back(0);
                for (int k = 0; k < speculate && !cand.empty(); k++) {
                    size_t bi = 0;
                    for (size_t c = 1; c < cand.size(); c++) if (h[cand[c]].lb < h[cand[bi]].lb) bi = c;
                    const size_t i = cand[bi];
                    cand[bi] = cand.back(); cand.pop_back();
                    if (2 * i + 1 < h.size()) cand.push_back(2 * i + 1);
                    if (2 * i + 2 < h.size()) cand.push_back(2 * i + 2);
                    if (optError - h[i].lb < sseThresh) continue;          // the search would stop at this cube
                    bool known = false;
                    for (const auto& w : ahead_of_time) known = known || same_cube(w->parent, h[i]);
                    if (known) continue;
                    if (ahead_of_time.size() >= (size_t)(4 * speculate)) { ahead_of_time.erase(ahead_of_time.begin()); spec_misses++; }
                    ahead_of_time.emplace_back(new Work);
                    prepare(h[i], *ahead_of_time.back());
                    for (Search& s : ahead_of_time.back()->ss) run.push_back(&s);
                    fresh.push_back(ahead_of_time.back().get());
                }
            }
            if (!run.empty() && comm.world > 1 && shard_cubes) {
                // Several ranks: cube i of the step belongs to rank i mod world, which runs its searches to the end on its own
                // GPU; then one sum over the ranks of a buffer with six slots per search (best error, whether a best cube was
                // found, that cube), zero in every slot a rank does not own: floats widened to double, added to zeros, exact.
                std::vector<Search*> mine;
                for (size_t i = 0; i < fresh.size(); i++)
                    if ((int)(i % (size_t)comm.world) == comm.rank)
                        for (Search& s : fresh[i]->ss) mine.push_back(&s);
                if (!mine.empty()) run_searches(mine);
                std::vector<double> buf(6 * run.size(), 0.0);
                size_t pos = 0;
                for (size_t i = 0; i < fresh.size(); i++)
                    for (Search& s : fresh[i]->ss) {
                        if ((int)(i % (size_t)comm.world) == comm.rank) {
                            double* b = &buf[6 * pos];
                            b[0] = s.optErrorT; b[1] = s.have_best ? 1.0 : 0.0;
                            if (s.have_best) { b[2] = s.best.x; b[3] = s.best.y; b[4] = s.best.z; b[5] = s.best.w; }
                        }
                        pos++;
                    }
                sum_over_ranks(buf);
                pos = 0;
                for (size_t i = 0; i < fresh.size(); i++)
                    for (Search& s : fresh[i]->ss) {
                        if ((int)(i % (size_t)comm.world) != comm.rank) {
                            const double* b = &buf[6 * pos];
                            s.optErrorT = (float)b[0]; s.have_best = b[1] != 0.0;
                            s.best.x = (float)b[2]; s.best.y = (float)b[3]; s.best.z = (float)b[4]; s.best.w = (float)b[5];
                            s.best.lb = s.best.ub = 0;
                            s.done = true;
                        }
                        pos++;
                    }
            } else if (!run.empty()) run_searches(run);
        } else {
            cur.reset(new Work);
            prepare(parent, *cur);
        }
        const std::vector<Child>& kids = cur->kids;
        std::vector<Search>& ss = cur->ss;

        bool finished = false;
        for (size_t k = 0; k < kids.size() && !finished; k++) {
            RotNode node = kids[k].node;
            const float* R = kids[k].R;
            std::memcpy(R_, R, sizeof R_);
            TransNode nodeTrans = InitTrans;
            float ub;
            if (batched) { ub = ss[2 * k].optErrorT; if (ss[2 * k].have_best) nodeTrans = ss[2 * k].best; }
            else ub = innerBnB(nullptr, &nodeTrans);
            if (ub < optError) {
                optError = ub;
                optNodeRot = node;
                optNodeTrans = nodeTrans;
                optMat = Matrix4d::Identity();
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) optMat(r, c) = R[3 * r + c];
                optMat(0, 3) = optNodeTrans.x + optNodeTrans.w / 2;
                optMat(1, 3) = optNodeTrans.y + optNodeTrans.w / 2;
                optMat(2, 3) = optNodeTrans.z + optNodeTrans.w / 2;
                tmpMat = optMat;
                { Stopwatch w_{g_t_icp}; error = runICP(tmpMat); }
                if (error < optError) { optError = error; optMat = tmpMat; }
                // drop queued rotation cubes that can no longer improve (src/GoIcp.cpp:317-328)
                RotQueue keep;
                while (!rq.empty()) {
                    const RotNode n = rq.top(); rq.pop();
                    if (n.lb < optError) keep.push(n); else break;
                }
                rq = keep;
            }
            const int lvl = std::min(node.l, MAXROTLEVEL - 1);
            float lb;
            if (batched) lb = ss[2 * k + 1].optErrorT;
            else lb = innerBnB(maxRotDist[lvl], nullptr);
            if (batched && lb > optError) lb = optError;   // a search started before optError improved
            if (lb >= optError) continue;
            node.lb = lb; node.ub = ub;
            if (optError - parent.lb < sseThresh) { finished = true; break; }
            rq.push(node);
        }
        if (finished) break;
    }
    spec_misses += (long long)ahead_of_time.size();
    if (comm.world > 1 && shard_cubes && batched) {   // bound evaluations and launches of the whole job, not of this rank's share
        std::vector<double> tot{(double)inner_evals, (double)launches};
        sum_over_ranks(tot);
        inner_evals = (long long)tot[0]; launches = (long long)tot[1];
    }
    if (std::getenv("ICPX_GOICP_VERBOSE"))
        std::fprintf(stderr, "GoIcp: %lld launches, %lld bound evaluations, %.3f s inside the batches; build %.3f take %.3f icp %.3f\n", launches, inner_evals, eval_seconds, g_t_build, g_t_take, g_t_icp), g_t_build = g_t_take = g_t_icp = 0;
    if (std::getenv("ICPX_GOICP_VERBOSE")) {
        std::fprintf(stderr, "GoIcp: %lld sets of searches, longest search of a set: %lld cubes summed over the sets, %lld at most; searches by cubes popped (<=1, 2, <=4, ...):", g_run_calls, g_longest_sum, g_longest_max);
        for (long long& h : g_pops_hist) { std::fprintf(stderr, " %lld", h); h = 0; }
        std::fprintf(stderr, "\n");
        g_run_calls = g_longest_sum = g_longest_max = 0;
    }
    return optError;
}

ICP_res GoIcp::registration() {
    Init();
    OuterBnB();
    clear();
    ICP_res ans;
    ans.resN3.data.resize((size_t)Nd * 3);
    for (int i = 0; i < Nd; i++) {
        const Point3f& s = pData->points[i];
        const double p[3] = {(double)s.x, (double)s.y, (double)s.z};
        apply4(optMat, p, &ans.resN3.data[3 * (size_t)i]);
    }
    ans.transMat = optMat;   // the reference leaves this field unset (src/GoIcp.cpp:542-547)
    return ans;
}

}  // namespace icp

// =================================================================================================
// extern "C" driver surface over the classes, for Python (tests, bench) — handles are opaque.
// =================================================================================================
using namespace icp;

namespace {
thread_local std::string g_err;
template <class F> int guard(F&& f) {
    try { f(); return 0; }
    catch (const IcpxFailure& e) { g_err = e.what(); return e.code; }
    catch (const std::exception& e) { g_err = e.what(); return -100; }
}
PointCloudTPtr make_cloud(const float* xyz, size_t n) {
    auto c = std::make_shared<PointCloudT>();
    c->points.resize(n);
    for (size_t i = 0; i < n; i++) c->points[i] = Point3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return c;
}
struct HCollective {
    void (*allreduce_host)(double*, int) = nullptr;
    void (*allreduce_dev)(double*, int) = nullptr;
    static void tramp(double* p, int n, void* user) { ((HCollective*)user)->allreduce_host(p, n); }
    static void tramp_dev(double* p, int n, void* user) { ((HCollective*)user)->allreduce_dev(p, n); }
};
struct HIcp { DevicePtr dev; PointCloudTPtr src, tgt; std::unique_ptr<SingleThreadIcp> icp; HCollective hc; };
struct HGo { DevicePtr dev; PointCloudTPtr model, data; std::unique_ptr<GoIcp> go; HCollective hc; };
struct HLdt { DevicePtr dev; PointCloudTPtr cloud; std::unique_ptr<LDT> ldt; };
struct HKd { DevicePtr dev; PointCloudTPtr cloud; std::unique_ptr<KDTree> kd; };
}  // namespace

extern "C" {

const char* icpxh_last_error() { return g_err.c_str(); }

void icpxh_svd3(const double* A9, double* U9, double* s3, double* V9) {
    Matrix3d A, U, V;
    std::memcpy(A.m, A9, sizeof A.m);
    svd3(A, U, s3, V);
    std::memcpy(U9, U.m, sizeof U.m); std::memcpy(V9, V.m, sizeof V.m);
}

void icpxh_kabsch(const double* H9, double* R9) {
    Matrix3d H;
    std::memcpy(H.m, H9, sizeof H.m);
    const Matrix3d R = kabsch_rotation(H);
    std::memcpy(R9, R.m, sizeof R.m);
}

// ---- cloud input
// Reads a PLY file; returns the number of valid vertices (or -1) and copies up to `cap` of them to xyz_out.
long icpxh_load_ply(const char* path, float* xyz_out, long cap) {
    PointCloudT c;
    std::string err;
    const long n = loadFile(path, c, &err);
    if (n < 0) { g_err = err; return -1; }
    for (long i = 0; i < n && i < cap; i++) { xyz_out[3 * i] = c.points[i].x; xyz_out[3 * i + 1] = c.points[i].y; xyz_out[3 * i + 2] = c.points[i].z; }
    return n;
}
// In-place Go-ICP normalisation of two packed xyz arrays; returns the scale.
double icpxh_normalize_goicp(float* model, size_t nm, float* data, size_t nd) {
    PointCloudTPtr m = make_cloud(model, nm), d = make_cloud(data, nd);
    const double s = normalize_goicp(*m, *d);
    for (size_t i = 0; i < nm; i++) { model[3 * i] = m->points[i].x; model[3 * i + 1] = m->points[i].y; model[3 * i + 2] = m->points[i].z; }
    for (size_t i = 0; i < nd; i++) { data[3 * i] = d->points[i].x; data[3 * i + 1] = d->points[i].y; data[3 * i + 2] = d->points[i].z; }
    return s;
}

// ---- KDTree
void* icpxh_kdtree_new(int device, const float* tgt, size_t n) {
    HKd* h = new HKd;
    if (guard([&] { h->dev = std::make_shared<Device>(device); h->cloud = make_cloud(tgt, n); h->kd.reset(new KDTree(h->cloud, 3, h->dev)); })) { delete h; return nullptr; }
    return h;
}
void icpxh_kdtree_free(void* p) { delete (HKd*)p; }
int icpxh_kdtree_get_closest(void* p, const double* q, size_t n, int32_t* idx, double* d2) {
    return guard([&] { HKd* h = (HKd*)p; if (n == 1) { Vector3d v; v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; idx[0] = h->kd->get_closest(v); if (d2) h->kd->get_closest(q, 1, idx, d2); } else h->kd->get_closest(q, n, idx, d2); });
}
int icpxh_kdtree_k_close(void* p, const double* q, size_t n, int k, int32_t* idx, double* d2) {
    return guard([&] { ((HKd*)p)->kd->k_close(q, n, k, idx, d2); });
}
int icpxh_kdtree_distance(void* p, const float* q, size_t n, float* out) {
    return guard([&] { HKd* h = (HKd*)p; if (n == 1) out[0] = h->kd->Distance(q[0], q[1], q[2]); else h->kd->Distance(q, n, out); });
}

// ---- the icp/ fork's loop
int icpxh_knn_icp(int device, const float* A, size_t na, const float* B, size_t nb, int max_iteration, float tolerance, int K, int remain,
                  double* trans16, float* distances, int* iter) {
    return guard([&] {
        const ICP_OUT r = knn_icp(make_cloud(A, na), make_cloud(B, nb), max_iteration, tolerance, K, remain, std::make_shared<Device>(device));
        std::memcpy(trans16, r.trans.m, 16 * sizeof(double));
        if (distances) std::memcpy(distances, r.distances.data(), r.distances.size() * sizeof(float));
        *iter = r.iter;
    });
}

// ---- LDT
void* icpxh_ldt_new(int device, const float* tgt, size_t n, uint32_t div) {
    HLdt* h = new HLdt;
    if (guard([&] { h->dev = std::make_shared<Device>(device); h->cloud = make_cloud(tgt, n); h->ldt.reset(new LDT(h->cloud, div, 2, h->dev)); })) { delete h; return nullptr; }
    return h;
}
void icpxh_ldt_free(void* p) { delete (HLdt*)p; }
int icpxh_ldt_info(void* p, float* min3, float* cellLen, uint32_t* div, uint64_t* occupied) {
    HLdt* h = (HLdt*)p;
    min3[0] = h->ldt->_min.x; min3[1] = h->ldt->_min.y; min3[2] = h->ldt->_min.z;
    *cellLen = h->ldt->cellLen; *div = h->ldt->div; *occupied = h->ldt->occupied;
    return 0;
}
int icpxh_ldt_distance(void* p, const float* q, size_t n, float* out) {
    return guard([&] { HLdt* h = (HLdt*)p; if (n == 1) out[0] = h->ldt->Distance(q[0], q[1], q[2]); else h->ldt->Distance(q, n, out); });
}
int icpxh_ldt_grid(void* p, float* g_out, int32_t* sq_out) {
    return guard([&] {
        HLdt* h = (HLdt*)p;
        if (sq_out) std::memcpy(sq_out, h->ldt->squared().data(), h->ldt->squared().size() * sizeof(int32_t));
        if (g_out) std::memcpy(g_out, h->ldt->grid().data(), h->ldt->grid().size() * sizeof(float));
    });
}

// ---- SingleThreadIcp
void* icpxh_icp_new(int device, const float* src, size_t ns, const float* tgt, size_t nt, int iters) {
    HIcp* h = new HIcp;
    if (guard([&] {
            h->dev = std::make_shared<Device>(device);
            h->src = make_cloud(src, ns); h->tgt = make_cloud(tgt, nt);
            h->icp.reset(new SingleThreadIcp(h->src, h->tgt, iters, h->dev));
        })) { delete h; return nullptr; }
    return h;
}
void icpxh_icp_free(void* p) { delete (HIcp*)p; }
int icpxh_icp_device_loop(void* p, int on) { ((HIcp*)p)->icp->device_loop = on != 0; return 0; }
// world > 1: this handle holds one slice of the source; allreduce sums `n` host doubles over the ranks in place.
int icpxh_icp_set_collective(void* p, int rank, int world, void (*allreduce)(double*, int)) {
    return guard([&] {
        HIcp* h = (HIcp*)p;
        h->hc.allreduce_host = allreduce;
        h->icp->comm.rank = rank; h->icp->comm.world = world; h->icp->comm.user = &h->hc;
        h->icp->comm.allreduce_sum_host = &HCollective::tramp;
        h->icp->setSource(h->src);   // refresh Nd_global
    });
}
// guess16 / T_out16 row-major; per_iter_T (iters x 16), mse (iters: the NN residual before each update), mse_after (iters: the
// residual the reference logs after each update) and resN3 (ns x 3) may be NULL.
int icpxh_icp_registration(void* p, const double* guess16, double* T_out16, double* per_iter_T, double* mse, double* resN3, int32_t* idx_last,
                           double* mse_after) {
    return guard([&] {
        HIcp* h = (HIcp*)p;
        Matrix4d g;
        std::memcpy(g.m, guess16, sizeof g.m);
        h->icp->keep_correspondences = idx_last != nullptr;
        ICP_res r = h->icp->registration(g, h->src);
        std::memcpy(T_out16, r.transMat.m, sizeof r.transMat.m);
        const int it = h->icp->max_iter;
        if (per_iter_T) for (int k = 0; k < it; k++) std::memcpy(per_iter_T + 16 * k, h->icp->T_trace[k].m, 16 * sizeof(double));
        if (mse) for (int k = 0; k < it; k++) mse[k] = h->icp->nn_mse_trace[k];
        if (mse_after) for (int k = 0; k < it; k++) mse_after[k] = h->icp->mse_trace[k];
        if (resN3) std::memcpy(resN3, r.resN3.data.data(), r.resN3.data.size() * sizeof(double));
        if (idx_last) std::memcpy(idx_last, h->icp->Neighbour.pointsPARA.data(), sizeof(int32_t) * h->icp->Nd);
    });
}

// ---- DT3D
struct HDt3d { DevicePtr dev; std::unique_ptr<DT3D> dt; };
// xyz: n x 3 doubles, row-major. info8 receives SIZE, scale, xMin, yMin, zMin, xMax, yMax, zMax.
void* icpxh_dt3d_new(int device, const double* xyz, size_t n, double expandFactor, double* info8) {
    HDt3d* h = new HDt3d;
    if (guard([&] {
            h->dev = std::make_shared<Device>(device);
            h->dt.reset(new DT3D(h->dev));
            h->dt->expandFactor = expandFactor;
            std::vector<double> x(n), y(n), z(n);
            for (size_t i = 0; i < n; i++) { x[i] = xyz[3 * i]; y[i] = xyz[3 * i + 1]; z[i] = xyz[3 * i + 2]; }
            h->dt->Build(x.data(), y.data(), z.data(), (int)n);
            const DT3D& d = *h->dt;
            const double v[8] = {(double)d.SIZE, d.scale, d.xMin, d.yMin, d.zMin, d.xMax, d.yMax, d.zMax};
            std::memcpy(info8, v, sizeof v);
        })) { delete h; return nullptr; }
    return h;
}
void icpxh_dt3d_free(void* p) { delete (HDt3d*)p; }
int icpxh_dt3d_distance(void* p, const double* q, size_t n, float* out) {
    return guard([&] { HDt3d* h = (HDt3d*)p; if (n == 1) out[0] = h->dt->Distance(q[0], q[1], q[2]); else h->dt->Distance(q, n, out); });
}

// ---- ICP3D
struct HIcp3d { DevicePtr dev; std::unique_ptr<ICP3D> icp; };
void* icpxh_icp3d_new(int device, const double* model, size_t m) {
    HIcp3d* h = new HIcp3d;
    if (guard([&] { h->dev = std::make_shared<Device>(device); h->icp.reset(new ICP3D(h->dev)); h->icp->Build(model, m); })) { delete h; return nullptr; }
    return h;
}
void icpxh_icp3d_free(void* p) { delete (HIcp3d*)p; }
// R9 / t3 in-out, row-major; do_trim / trim_fraction are the public fields of the class. Returns the error through *err.
int icpxh_icp3d_run(void* p, const double* data, size_t n, double* R9, double* t3, size_t max_iter, double err_diff, int do_trim,
                    double trim_fraction, double* err, size_t* iterations) {
    return guard([&] {
        HIcp3d* h = (HIcp3d*)p;
        h->icp->do_trim = do_trim != 0; h->icp->trim_fraction = trim_fraction;
        Matrix3d R; Vector3d t;
        std::memcpy(R.m, R9, sizeof R.m);
        for (int i = 0; i < 3; i++) t[i] = t3[i];
        *err = h->icp->Run(data, n, R, t, max_iter, err_diff);
        std::memcpy(R9, R.m, sizeof R.m);
        for (int i = 0; i < 3; i++) t3[i] = t[i];
        if (iterations) *iterations = h->icp->iterations;
    });
}

// ---- GoIcp
void* icpxh_goicp_new(int device, const float* model, size_t nm, const float* data, size_t nd, double mse, uint32_t dt_div) {
    HGo* h = new HGo;
    if (guard([&] {
            h->dev = std::make_shared<Device>(device);
            h->model = make_cloud(model, nm); h->data = make_cloud(data, nd);
            h->go.reset(new GoIcp(h->model, h->data, 10000, mse, h->dev, dt_div));
        })) { delete h; return nullptr; }
    return h;
}
void icpxh_goicp_free(void* p) { delete (HGo*)p; }
int icpxh_goicp_options(void* p, int run_bnb, int batched, int parity_sums, int frontier_width, long long max_rot_pops, int lookahead,
                        int speculate) {
    HGo* h = (HGo*)p;
    h->go->run_bnb = run_bnb != 0; h->go->batched = batched != 0; h->go->parity_sums = parity_sums != 0;
    h->go->frontier_width = frontier_width; h->go->max_rot_pops = max_rot_pops; h->go->lookahead = lookahead;
    h->go->speculate = speculate;
    return 0;
}
int icpxh_goicp_exact_residuals(void* p, int on) { ((HGo*)p)->go->exact_residuals = on != 0; return 0; }
int icpxh_goicp_shard_cubes(void* p, int on) { ((HGo*)p)->go->shard_cubes = on != 0; return 0; }
int icpxh_goicp_pipeline(void* p, int on) { ((HGo*)p)->go->pipeline = on != 0; return 0; }
int icpxh_goicp_tail_lookahead(void* p, int on) { ((HGo*)p)->go->tail_lookahead = on != 0; return 0; }
int icpxh_goicp_device_search(void* p, int on) { ((HGo*)p)->go->device_search = on != 0; return 0; }
// Device-side collective: allreduce_dev sums n doubles at a device address inside [dev_buffer, dev_buffer + capacity) over
// the ranks, in place, on the stream given here (which the context then uses for everything).
int icpxh_goicp_set_collective_dev(void* p, int rank, int world, void (*allreduce_dev)(double*, int), double* dev_buffer,
                                   size_t capacity, void* hip_stream) {
    return guard([&] {
        HGo* h = (HGo*)p;
        h->hc.allreduce_dev = allreduce_dev;
        h->go->comm.rank = rank; h->go->comm.world = world; h->go->comm.user = &h->hc;
        h->go->comm.allreduce_sum_dev = &HCollective::tramp_dev;
        h->go->comm.dev_buffer = dev_buffer; h->go->comm.dev_capacity = capacity;
        h->dev->check(icpx_set_stream(h->dev->ctx(), hip_stream));
    });
}
int icpxh_goicp_set_collective(void* p, int rank, int world, void (*allreduce)(double*, int)) {
    HGo* h = (HGo*)p;
    h->hc.allreduce_host = allreduce;
    h->go->comm.rank = rank; h->go->comm.world = world; h->go->comm.user = &h->hc;
    h->go->comm.allreduce_sum_host = &HCollective::tramp;
    return 0;
}
// stats: optError, initError, icpError, rot_pops, inner_evals, final_lb, launches, sseThresh, spec_hits, spec_misses, collectives
int icpxh_goicp_registration(void* p, double* optMat16, double* stats8, double* resN3) {
    return guard([&] {
        HGo* h = (HGo*)p;
        ICP_res r = h->go->registration();
        std::memcpy(optMat16, h->go->optMat.m, 16 * sizeof(double));
        stats8[0] = h->go->optError; stats8[1] = h->go->initError; stats8[2] = h->go->icpError;
        stats8[3] = (double)h->go->rot_pops; stats8[4] = (double)h->go->inner_evals; stats8[5] = h->go->final_lb;
        stats8[6] = (double)h->go->launches; stats8[7] = h->go->sseThresh;
        stats8[8] = (double)h->go->spec_hits; stats8[9] = (double)h->go->spec_misses; stats8[10] = (double)h->go->collectives; stats8[11] = (double)h->go->search_steps;
        if (resN3) std::memcpy(resN3, r.resN3.data.data(), r.resN3.data.size() * sizeof(double));
    });
}
// A single innerBnB call under rotation R9 with the given optError (exposes the member for parity tests).
int icpxh_goicp_inner(void* p, const float* R9, int level, float optError, float* result, float* trans4) {
    return guard([&] {
        HGo* h = (HGo*)p;
        GoIcp& g = *h->go;
        if (!g.maxRotDist) g.Init();
        g.optError = optError;
        TransNode t = g.InitTrans;
        g.setRotation(R9);
        *result = g.innerBnB(level < 0 ? nullptr : g.maxRotDist[level], &t);
        trans4[0] = t.x; trans4[1] = t.y; trans4[2] = t.z; trans4[3] = t.w;
    });
}

}  // extern "C"
