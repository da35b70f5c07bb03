// C++ host classes with the shapes of the reference's registration API, running on the icpx C ABI.
//
//   icp::Registration / icp::ICP_res   src/Registration.hpp:10-24
//   icp::KDTree                        src/kdtree.cpp:30-152        (get_closest, Distance)
//   icp::LDT                           src/testLDT.cpp:26-37        (Distance; public g, _min, cellLen, div)
//   icp::SingleThreadIcp               src/SingleThreadIcp.cpp:46-94
//   icp::GoIcp                         src/GoIcp.cpp:44-79
//
// PCL, Eigen and boost are not available, so the cloud and matrix types are small stand-alone structs
// with the members the reference's callers touch (points, x/y/z, operator()(r,c), rows()).
// Only include/icpx.h is used underneath: no HIP, no torch.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "icpx.h"

namespace icp {

struct Point3f {            // 16-byte point like pcl::PointXYZ (src/common.hpp:28)
    float x = 0, y = 0, z = 0, pad = 1.0f;
    Point3f() = default;
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct PointCloudT {
    std::vector<Point3f> points;
    size_t size() const { return points.size(); }
};
using PointCloudTPtr = std::shared_ptr<PointCloudT>;

struct Vector3d {
    double v[3] = {0, 0, 0};
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};

struct Matrix3d {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};

struct Matrix4d {           // row-major storage, (row, col) access like Eigen
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double& operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
    static Matrix4d Identity() { return Matrix4d(); }
    Matrix4d operator*(const Matrix4d& o) const;
};

struct MatrixN3 {           // N x 3 doubles, row-major (stands in for Eigen::MatrixXd resN3)
    std::vector<double> data;
    size_t rows() const { return data.size() / 3; }
    size_t cols() const { return 3; }
    double& operator()(size_t r, size_t c) { return data[3 * r + c]; }
    double operator()(size_t r, size_t c) const { return data[3 * r + c]; }
};

struct ICP_res {            // src/Registration.hpp:10-14
    MatrixN3 resN3;
    Matrix4d transMat;
};

class Registration {        // src/Registration.hpp:17-24
public:
    virtual ~Registration() {}
    virtual void setTarget(PointCloudTPtr _target) = 0;
    virtual ICP_res registration(Matrix4d&) = 0;
};

struct IcpxFailure : std::runtime_error {
    int code;
    IcpxFailure(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// Shared handle on one icpx context (one GPU). Classes built on the same Device share the target
// cloud, its search grid and its distance transform.
class Device {
public:
    explicit Device(int device = 0);
    ~Device();
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    icpx_ctx* ctx() const { return ctx_; }
    void check(int rc) const;   // throws IcpxFailure with icpx_last_error
    // the cloud last given to icpx_set_target: the object (weak, so a dead cloud's address is never mistaken for a new
    // cloud's), its size and its storage
    std::weak_ptr<struct PointCloudT> target_ref;
    size_t target_n = 0;
    const void* target_first = nullptr;
    void forget_target() { target_ref.reset(); target_n = 0; target_first = nullptr; edt_div = 0; edt_owner = nullptr; }
    uint32_t edt_div = 0;               // resolution of the EDT built for that target by LDT's box rule (0 = none)
    const void* edt_owner = nullptr;    // the DT3D object whose grid the context holds instead (nullptr: none)
private:
    icpx_ctx* ctx_ = nullptr;
};
using DevicePtr = std::shared_ptr<Device>;

// Optional hooks that make the host loops collective over several processes (one per GPU). Each
// process holds a slice of the source points (ICP) or evaluates a slice of every bound batch (Go-ICP).
struct Collective {
    int rank = 0, world = 1;
    void* user = nullptr;
    // in-place sum over ranks of n doubles that live in DEVICE memory of this rank's GPU, enqueued on the stream the
    // rank's icpx context uses (RCCL through the caller's communicator; icpx_set_stream makes the two streams one)
    void (*allreduce_sum_dev)(double* dev, int n, void* user) = nullptr;
    // device memory of this rank's GPU that the classes may use for what they hand to allreduce_sum_dev
    double* dev_buffer = nullptr;
    size_t dev_capacity = 0;      // in doubles
    // in-place sum over ranks of n doubles in HOST memory
    void (*allreduce_sum_host)(double* host, int n, void* user) = nullptr;
};

// 3x3 helpers (host side of best_fit_transform, src/SingleThreadIcp.cpp:249-272)
void svd3(const Matrix3d& A, Matrix3d& U, double sig[3], Matrix3d& V);
Matrix3d kabsch_rotation(const Matrix3d& H);

class KDTree {              // src/kdtree.cpp:30-152
public:
    KDTree(const PointCloudTPtr& cloudptr, int _dims = 3, DevicePtr dev = nullptr);
    int get_closest(const Vector3d& x);                       // src/kdtree.cpp:107-122
    // src/kdtree.cpp:124-150 fills a priority queue with the k nearest; here they come back as indices and
    // squared distances, nearest first (n queries, row-major n x 3; outputs n x k)
    void k_close(const double* q, size_t n, int k, int32_t* idx, double* d2 = nullptr);
    float Distance(float x0, float x1, float x2);             // src/kdtree.cpp:91-105
    // batch forms of the two calls above (n queries, row-major n x 3)
    void get_closest(const double* q, size_t n, int32_t* idx, double* d2 = nullptr);
    void Distance(const float* q, size_t n, float* out);
    int dims;
    PointCloudTPtr target;
    DevicePtr dev;
};

// The loop of the reference's icp/ fork (icp/src/icp.hpp:398-460) around its k-nearest-neighbour search (:310-359) and its
// best_fit_transform over K neighbours (:185-253), restated as a function of its input. The fork qsorts a vector of
// structs that own std::vectors with a comparator that truncates a float difference to int, so what it computes is not
// determined by its input; here the order is that comparator applied by a stable bottom-up merge sort (with squared
// distances below 1 it leaves the source order), the K neighbours are the exact nearest by (distance, index), and the
// arithmetic follows the fork statement by statement: the running "centroid_B /= K" inside the point loop (:205), the
// row of BB that keeps only the last neighbour (:216-221), the rows beyond `num` that stay copies of A (:192-193,
// remainPercentage < 100), R = U V^T with the sign fix on V's third column (:240-246). Needs nb <= na, as the fork
// does (it indexes a matrix of min(na, nb) target points with neighbours found among all nb).
struct ICP_OUT {            // icp/src/icp.hpp:23-27
    Matrix4d trans;
    std::vector<float> distances;   // mean squared distance to the K neighbours per source point, in the sorted order of the last iteration
    int iter = 0;
};
ICP_OUT knn_icp(const PointCloudTPtr& A, const PointCloudTPtr& B, int max_iteration, float tolerance, int Ksearch = 5,
                int remainPercentage = 100, DevicePtr dev = nullptr);

class LDT {                 // src/testLDT.cpp:26-37
public:
    LDT(const PointCloudTPtr&, uint32_t _div = 300, int _core = 2, DevicePtr dev = nullptr);
    float Distance(float, float, float);                      // src/testLDT.cpp:235-290
    void Distance(const float* q, size_t n, float* out);      // batch form
    const std::vector<int32_t>& squared();                    // the int32 squared grid, [x][y][z]
    const std::vector<float>& grid();                         // g = sqrt(squared), as the reference stores it
    Point3f _min, _max;
    uint32_t div;
    float cellLen;
    int CORES;                                                // kept for signature parity; unused
    uint64_t occupied = 0;
    DevicePtr dev;
private:
    PointCloudTPtr cloud_;
    std::vector<int32_t> sq_;
    std::vector<float> g_;
};

// The distance-transform class of the original Go-ICP code that the reference carries (src/jly_3ddt.h:37-52,
// src/jly_3ddt.cpp:727-856): cubic grid of SIZE = 300 cells around the cloud's bounding box enlarged by
// expandFactor, rounded voxel indices. The grid holds exact distances (the reference's DEuclidean pass is
// an approximation of them).
class DT3D {
public:
    explicit DT3D(DevicePtr dev = nullptr);
    void Build(const double* x, const double* y, const double* z, int num);   // :731-819
    float Distance(double x, double y, double z);                              // :821-856
    void Distance(const double* xyz, size_t n, float* out);                    // batch form, n x 3 row-major
    int SIZE;
    double scale, expandFactor;
    double xMin, xMax, yMin, yMax, zMin, zMax;
    DevicePtr dev;
private:
    PointCloudTPtr cloud_;
    void ensure_grid();
};

struct Correspondance {     // src/SingleThreadIcp.cpp:37-44
    std::vector<int> pointsPARA;
    Matrix3d covariance;
};

class SingleThreadIcp {     // src/SingleThreadIcp.cpp:46-94
public:
    SingleThreadIcp(PointCloudTPtr _s, PointCloudTPtr _t, int _iter = 10, DevicePtr dev = nullptr);
    virtual ~SingleThreadIcp() {}
    void setTarget(PointCloudTPtr _target);
    void setSource(const PointCloudTPtr _source);
    virtual ICP_res registration(Matrix4d&, const PointCloudTPtr&);

    void kdtree_search();          // transform + NN + sums on the GPU (src/SingleThreadIcp.cpp:186-214)
    void cal_covarance();          // means and H from the sums (:216-247)
    Matrix4d& best_fit_transform();   // SVD -> T, accumulates finMat (:249-290)

    PointCloudTPtr source, target;
    int max_iter;
    Correspondance Neighbour;
    int Nd, Nm;
    Matrix4d T;                    // last per-iteration transform
    Matrix4d finMat;               // accumulated transform applied to the source on the GPU
    Vector3d mean_tar, mean_tmpdata;
    // registration() as one call of icpx_icp_run: the iterations follow each other on the GPU, the small dense part (means,
    // covariance, SVD, update) done by one thread there with the arithmetic of cal_covarance / best_fit_transform (the same
    // source compiled for both sides). One process, no correspondences wanted; mean_tmpdata and Neighbour.covariance, the
    // last iteration's intermediates, are then not filled. false: the loop below, a round trip per iteration.
    bool device_loop = true;
    bool keep_correspondences = false;   // fill Neighbour.pointsPARA every iteration (costs a device->host copy)
    std::vector<double> mse_trace;       // per iteration, sum |T p - q|^2 / Nd AFTER the update with the pairs it was fitted to: the value the reference logs (src/SingleThreadIcp.cpp:281-288)
    std::vector<double> nn_mse_trace;    // per iteration, sum |p - q|^2 / Nd BEFORE the update (the NN residual)
    std::vector<Matrix4d> T_trace;       // per-iteration T
    Collective comm;                     // world > 1: this object holds one slice of the source
    long long Nd_global = 0;             // total source points over all ranks (= Nd when world == 1)
    DevicePtr dev;
private:
    double sums_[ICPX_NSUMS];
    void ensure_target();
};

// The ICP of the original Go-ICP code base that the reference carries but does not call
// (src/jly_icp3d.cpp:18-60, :181-295): nanoflann NN, optional trimming to the best (1 - trim_fraction) of
// the pairs, stop when the summed squared error improves by less than err_diff per kept pair.
// T = double only; the clouds are held as float32 on the GPU, so coordinates must be float-representable
// for parity with the reference (they are for every PLY it ships).
class ICP3D {
public:
    explicit ICP3D(DevicePtr dev = nullptr);
    void Build(const double* model, size_t n);                                       // :130-154
    double Run(const double* data, size_t n, Matrix3d& R, Vector3d& t);              // :162-166
    double Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, size_t max_iter);
    double Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, double err_diff);
    double Run(const double* data, size_t n, Matrix3d& R, Vector3d& t, size_t max_iter, double err_diff);   // :181-295
    double err_diff_def;      // 0.000001
    size_t max_iter_def;      // 10000
    double trim_fraction;     // 0
    bool do_trim;             // true
    size_t iterations = 0;    // passes of the last Run, the one that met the stop test included
    DevicePtr dev;
private:
    PointCloudTPtr model_, data_;
    std::vector<double> data_key_;
};

struct RotNode {            // src/GoIcp.cpp:25-33
    float a, b, c, w;
    double lb, ub;
    int l;
    friend bool operator<(const RotNode& n1, const RotNode& n2) { return n1.lb > n2.lb; }
};

struct TransNode {          // src/GoIcp.cpp:35-42
    float x, y, z, w;
    double lb, ub;
    friend bool operator<(const TransNode& n1, const TransNode& n2) { return n1.lb > n2.lb; }
};

class GoIcp {               // src/GoIcp.cpp:44-79
public:
    GoIcp(PointCloudTPtr pModel, PointCloudTPtr pData, int iter = 1e4, double _mse = 0.001, DevicePtr dev = nullptr,
          uint32_t dt_div = 300);
    ~GoIcp();
    void Init();                                   // src/GoIcp.cpp:459-489
    void clear();                                  // :529-535
    void setTarget(PointCloudTPtr _target) { pModel = _target; }
    ICP_res registration();                        // :537-548
    float runICP(Matrix4d&);                       // :106-139
    float innerBnB(float* maxDistL, TransNode* transOut);   // :353-457 (maxDistL must be NULL or maxRotDist[level])
    float OuterBnB();                              // :142-350 with the search after the initial ICP enabled
    // The reference's innerBnB reads the rotated cloud pDataTemp that OuterBnB prepared (:264-272); here
    // the rotation itself is the state, and the bound kernel rotates on the fly. R is row-major 3x3.
    void setRotation(const float R[9]) { for (int i = 0; i < 9; i++) R_[i] = R[i]; }

    PointCloudTPtr pModel, pData;
    int Nd, Nm;
    float optError;
    Matrix4d optMat;
    RotNode optNodeRot;
    TransNode optNodeTrans;
    float mseThresh, sseThresh;
    RotNode InitRot;
    TransNode InitTrans;
    float** maxRotDist = nullptr;                  // [ICPX_MAXROTLEVEL][Nd], host copy of the device table
    std::vector<float> minDis;
    SingleThreadIcp icp;
    int ITER;

    // Search controls that the reference does not have.
    bool run_bnb = true;          // false: stop after the initial ICP, like the committed early return (src/GoIcp.cpp:189)
    bool batched = true;          // true: the 2 x 8 inner searches of one rotation parent advance in lock-step, one launch per step
    bool parity_sums = false;     // true: bounds are the reference's sequential float sums of the per-point distances
    // The committed reference declares `// LDT dt; KDTree dt;` (src/GoIcp.cpp:76-77): its initial error (:164) and the SSE
    // after an ICP (:125) are sums of EXACT nearest-neighbour distances. true reproduces that (icpx_nn_distance); false, the
    // default, takes them from the distance transform like the bounds, which is what Go-ICP itself does and keeps errors
    // and bounds comparable when the rotation search runs (they differ from the exact ones by at most sqrt(3) cells per point).
    bool exact_residuals = false;
    // Translation parents popped per inner search and step (batched mode). The reference's inner search returns at
    // the first evaluated child whose lower bound is within sseThresh of the best error (src/GoIcp.cpp:445-449), so
    // the expansion order decides where it stops: 1 keeps the reference's order, larger values are faster per cube
    // but may end with a different (still threshold-accepted) result.
    int frontier_width = 1;
    // Translation cubes whose children are evaluated ahead of time (frontier_width 1 only): every step also sends the
    // children of the `lookahead` cubes nearest the top of each search's queue, and keeps their bounds until the
    // search pops those cubes. A bound depends on the cube alone, so the search takes exactly the decisions of the
    // reference's order, in fewer and larger launches; bounds of cubes that are never popped are wasted work.
    int lookahead = 4;            // bunny pair: 93 490 launches with 0, 29 634 with 4 (about 40 % less time), 24 285 with 16 (more time again)
    // Rotation cubes whose inner searches run ahead of time, in the same launches as those of the cube just popped
    // (batched mode): the `speculate` cubes that follow it in the rotation queue. Their searches start from the current
    // optError, so what they find is what the search would find later, as long as optError has not changed by then and
    // the cube is indeed popped next; otherwise it is thrown away. Decisions and result are those of speculate = 0; the
    // launches carry (1 + speculate) times as many cubes, which is what several GPUs sharing every launch need.
    int speculate = 0;
    long long spec_hits = 0, spec_misses = 0;
    // The searches of one step advance as two groups whose bound batches alternate on the GPU: the host digests one
    // group's bounds while the other group's batch is evaluated (one GPU, device sums). Decisions are unchanged: a
    // search never looks at another search's state.
    bool pipeline = true;
    bool tail_lookahead = true;   // the last few running searches of a step look further ahead (fewer, fuller batches; same decisions)
    // The inner searches of a step run on the device from start to end (icpx_bnb_search_run): queues, decisions and the
    // lists of cubes to bound stay there, the host enqueues steps and collects (best error, best cube) per search.
    // Used when frontier_width is 1, the sums are the device's (not parity_sums) and no bound batch is shared between
    // ranks; lookahead, tail_lookahead and pipeline then have nothing to do. Decisions are those of the host loop.
    bool device_search = true;
    long long search_steps = 0;   // steps the device searches of the last OuterBnB took (one launch each)
    long long max_rot_pops = 0;   // 0 = unlimited
    // world > 1, batched mode. true (default): the rotation cubes of a step (the popped one and the `speculate` searched
    // ahead with it) are dealt out to the ranks; each rank runs the complete inner searches of its cubes on its own GPU
    // without any communication, and ONE sum over the ranks (of a buffer in which every rank fills only the slots of
    // its own searches: best error, best cube) ends the step. A search depends on its cube and the starting error alone,
    // so every decision is that of one rank. false: every bound batch is cut into shares instead, one collective per
    // launch (what the first rounds did; kept for comparison).
    bool shard_cubes = true;
    Collective comm;              // world > 1: see shard_cubes
    long long collectives = 0;    // sums over the ranks that the last OuterBnB asked for
    // statistics
    long long rot_pops = 0, inner_evals = 0, launches = 0;
    double eval_seconds = 0;      // wall time spent inside the bound batches (launch, collective, copy back) of the last OuterBnB
    float initError = 0, icpError = 0;
    double final_lb = 0;
    DevicePtr dev;

private:
    struct Search;
    float R_[9];                                   // rotation of the current cube (innerBnB operates on it)
    void residuals(const Matrix4d& M, float* dist);
    int cur_level_of(const float* maxDistL) const;
    void eval_items(std::vector<icpx_bnb_item>& items, std::vector<float>& ub, std::vector<float>& lb);
    void run_searches(const std::vector<Search*>& ss);
    void run_searches_host(const std::vector<Search*>& ss);
    bool run_searches_device(const std::vector<Search*>& ss, std::vector<Search*>& left);
    void submit_parents(int slot, const std::vector<icpx_bnb_item>& parents);
    void collect_parents(int slot, std::vector<float>& ub, std::vector<float>& lb);
    bool shared_launches() const { return comm.world > 1 && !(shard_cubes && batched); }   // every launch cut into shares
    void sum_over_ranks(std::vector<double>& v);
    std::vector<float> slot_ub[2], slot_lb[2];   // bounds of a batch that was evaluated where it was submitted
    std::vector<double> slot_d;
    size_t slot_count[2] = {0, 0};
    bool slot_async[2] = {false, false};
};

}  // namespace icp
