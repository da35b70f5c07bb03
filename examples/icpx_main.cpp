// Command-line driver with the argument order of the reference's main (src/main.cpp:247-301):
//   icpx_main <source.ply> <target.ply> [iterations = 30] [mse = 0.001] [mode = icp | goicp | ldt]
// icp    test_icp    (src/main.cpp:64-73):   SingleThreadIcp(source, target, iterations).registration(I)
// goicp  test_goicp  (src/main.cpp:117-122): GoIcp(target, source, iterations, mse).registration() on the
//                    normalised clouds, with the rotation search enabled
// ldt    test_LinearDT (src/main.cpp:215-243): LDT(target, 300) against KDTree::Distance on the source
// There is no viewer; results are printed.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "icpx_host.hpp"
#include "icpx_io.hpp"

using namespace icp;
using Clock = std::chrono::steady_clock;

static void print_mat(const Matrix4d& M) {
    for (int r = 0; r < 4; r++) std::printf("  % .9f % .9f % .9f % .9f\n", M(r, 0), M(r, 1), M(r, 2), M(r, 3));
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s source.ply target.ply [iterations] [mse] [icp|goicp|ldt]\n", argv[0]); return 2; }
    const int ite = argc > 3 ? std::atoi(argv[3]) : 30;
    const double mse = argc > 4 ? std::atof(argv[4]) : 0.001;
    const char* mode = argc > 5 ? argv[5] : "icp";
    auto src = std::make_shared<PointCloudT>(), tgt = std::make_shared<PointCloudT>();
    std::string err;
    if (loadFile(argv[1], *src, &err) < 0 || loadFile(argv[2], *tgt, &err) < 0) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::printf("source %zu points, target %zu points\n", src->size(), tgt->size());
    try {
        auto dev = std::make_shared<Device>(0);
        const auto t0 = Clock::now();
        if (!std::strcmp(mode, "icp")) {
            SingleThreadIcp icp(src, tgt, ite, dev);
            Matrix4d guess = Matrix4d::Identity();
            ICP_res r = icp.registration(guess, src);
            std::printf("ICP, %d iterations, each transform mse %.6e -> %.6e\ntransMat:\n", ite, icp.mse_trace.front(), icp.mse_trace.back());
            print_mat(r.transMat);
        } else if (!std::strcmp(mode, "goicp")) {
            const double s = normalize_goicp(*tgt, *src);
            std::printf("normalised by %.6f\n", s);
            GoIcp go(tgt, src, ite, mse, dev);
            // the reference's expansion order (one translation cube per inner search and step); cubes evaluated and
            // rotation cubes searched ahead of time only fill the launches, every decision stays that of the plain order
            go.frontier_width = 1;
            go.lookahead = 1;
            go.speculate = 255;
            go.registration();
            std::printf("Go-ICP: init %.6f, after ICP %.6f, optimum %.6f (lower bound %.6f, epsilon %.6f), %lld rotation cubes, %lld bound evaluations\noptMat:\n",
                        go.initError, go.icpError, go.optError, go.final_lb, go.sseThresh, go.rot_pops, go.inner_evals);
            print_mat(go.optMat);
        } else if (!std::strcmp(mode, "ldt")) {
            LDT ldt(tgt, 300, 2, dev);
            KDTree kd(tgt, 3, dev);
            std::vector<float> q(src->size() * 3), a(src->size()), b(src->size());
            for (size_t i = 0; i < src->size(); i++) { q[3 * i] = src->points[i].x; q[3 * i + 1] = src->points[i].y; q[3 * i + 2] = src->points[i].z; }
            ldt.Distance(q.data(), src->size(), a.data());
            kd.Distance(q.data(), src->size(), b.data());
            double worst = 0;
            for (size_t i = 0; i < src->size(); i++) worst = std::fmax(worst, std::fabs((double)a[i] - b[i]));
            std::printf("LDT div %u cellLen %.9g occupied %llu; max |LDT - kd-tree| over the source = %.6g (sqrt3 * cellLen = %.6g)\n", ldt.div,
                        ldt.cellLen, (unsigned long long)ldt.occupied, worst, std::sqrt(3.0) * ldt.cellLen);
        } else { std::fprintf(stderr, "unknown mode %s\n", mode); return 2; }
        std::printf("elapsed %.3f s\n", std::chrono::duration<double>(Clock::now() - t0).count());
    } catch (const IcpxFailure& e) { std::fprintf(stderr, "icpx error %d: %s\n", e.code, e.what()); return 1; }
    return 0;
}
