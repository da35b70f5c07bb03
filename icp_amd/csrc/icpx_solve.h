// The small dense part of an ICP iteration — means and covariance from the sixteen sums, the 3 x 3 SVD, the rigid
// update (src/SingleThreadIcp.cpp:216-290) — written once for the host classes (g++) and for the kernel that runs a
// whole registration loop on the device (hipcc): the same statements in the same order, only + - * / sqrt and
// comparisons, no contraction (both are built with -ffp-contract=off), so both give the same bits.
// Plain row-major arrays: 3 x 3 as double[9], 4 x 4 as double[16].
#pragma once

#if defined(__HIPCC__)
#define ICPX_HD __host__ __device__ inline
#else
#include <cmath>
#define ICPX_HD inline
#endif

namespace icpx_solve {

ICPX_HD double sqrt_rn(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dsqrt_rn(x);
#else
    return std::sqrt(x);
#endif
}
ICPX_HD double abs_of(double x) { return x < 0 ? -x : x; }
ICPX_HD double max_of(double a, double b) { return a < b ? b : a; }   // std::max: the first unless it is smaller

ICPX_HD double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// One-sided (Hestenes) Jacobi SVD of a 3x3 matrix: columns of W = A V are rotated pairwise until they are orthogonal;
// singular values are their norms, descending. Accurate to working precision without forming A^T A.
ICPX_HD void svd3(const double A[9], double U[9], double sig[3], double V[9]) {
    double W[3][3], Vv[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { W[i][j] = A[3 * i + j]; Vv[i][j] = (i == j); }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; i++) { alpha += W[i][p] * W[i][p]; beta += W[i][q] * W[i][q]; gamma += W[i][p] * W[i][q]; }
                if (gamma == 0.0) continue;
                off = max_of(off, abs_of(gamma) / sqrt_rn(max_of(alpha * beta, 1e-300)));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (abs_of(zeta) + sqrt_rn(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt_rn(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; i++) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = c * wp - s * wq; W[i][q] = s * wp + c * wq;
                    const double vp = Vv[i][p], vq = Vv[i][q];
                    Vv[i][p] = c * vp - s * vq; Vv[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; j++) nrm[j] = sqrt_rn(W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j]);
    // descending by norm, equal norms in their order (an insertion sort, as std::sort does for three elements)
    for (int a = 1; a < 3; a++) {
        const int o = ord[a];
        int b = a;
        while (b > 0 && nrm[o] > nrm[ord[b - 1]]) { ord[b] = ord[b - 1]; b--; }
        ord[b] = o;
    }
    double u[3][3];
    for (int j = 0; j < 3; j++) {
        const int o = ord[j];
        sig[j] = nrm[o];
        for (int i = 0; i < 3; i++) { V[3 * i + j] = Vv[i][o]; u[j][i] = W[i][o]; }
    }
    // Normalise U; complete it when the input is rank deficient.
    const double tiny = 1e-14 * max_of(sig[0], 1e-300);
    if (sig[0] > 0) for (int i = 0; i < 3; i++) u[0][i] /= sig[0]; else { u[0][0] = 1; u[0][1] = u[0][2] = 0; }
    if (sig[1] > tiny) for (int i = 0; i < 3; i++) u[1][i] /= sig[1];
    else {
        int mI = 0;
        for (int i = 1; i < 3; i++) if (abs_of(u[0][i]) < abs_of(u[0][mI])) mI = i;
        double e[3] = {0, 0, 0}; e[mI] = 1;
        const double d = u[0][mI];
        double n = 0;
        for (int i = 0; i < 3; i++) { u[1][i] = e[i] - d * u[0][i]; n += u[1][i] * u[1][i]; }
        n = sqrt_rn(n);
        for (int i = 0; i < 3; i++) u[1][i] /= n;
    }
    const double cx[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2], u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    if (sig[2] > tiny) {
        const double d = cx[0] * u[2][0] + cx[1] * u[2][1] + cx[2] * u[2][2];
        for (int i = 0; i < 3; i++) u[2][i] = (d < 0 ? -cx[i] : cx[i]);
    } else for (int i = 0; i < 3; i++) u[2][i] = cx[i];
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) U[3 * i + j] = u[j][i];
}

// R = (U V^T)^T; when det R < 0 the last row of V^T is negated first (src/SingleThreadIcp.cpp:256-266).
ICPX_HD void kabsch_rotation(const double H[9], double R[9]) {
    double U[9], V[9], s[3];
    svd3(H, U, s, V);
    for (int pass = 0; pass < 2; pass++) {
        const double sgn = pass == 0 ? 1.0 : -1.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * j + 0] * V[3 * i + 0] + U[3 * j + 1] * V[3 * i + 1] + sgn * U[3 * j + 2] * V[3 * i + 2];
        if (!(det3(R) < 0)) break;
    }
}

// r = a * b for 4 x 4 matrices, every entry from a zero accumulator in k order.
ICPX_HD void mul4(const double a[16], const double b[16], double r[16]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0;
            for (int k = 0; k < 4; k++) acc += a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = acc;
        }
}

// Means and covariance from the sums of a pass {sum p (3), sum q (3), sum p q^T (9), sum |p - q|^2} over N pairs
// (src/SingleThreadIcp.cpp:216-247). mean_tar (in / out): the reference adds the matched target points onto the
// PREVIOUS mean before dividing (:229-231); reproduced, because it moves the result by ~1e-4. denom: (double)(float)N,
// the "/= (float)Nd" of :244. sum (p - pbar)(q - m)^T = sum p q^T - pbar (sum q)^T for any m, since sum (p - pbar) = 0.
ICPX_HD void covariance(const double sums[16], double N, double denom, double mean_tar[3], double mean_src[3], double H[9]) {
    for (int c = 0; c < 3; c++) mean_src[c] = sums[c] / N;
    for (int c = 0; c < 3; c++) mean_tar[c] = (mean_tar[c] + sums[3 + c]) / N;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[3 * r + c] = (sums[6 + 3 * r + c] - mean_src[r] * sums[3 + c]) / denom;
}

// The rigid update from the covariance (src/SingleThreadIcp.cpp:249-290): T = [R | mean_tar - R mean_src], and the
// residual the reference logs after applying T with the pairs it was fitted to (:281-288), from the same sums:
//   sum |Rp + t - q|^2 = sum |p - q|^2 + 2 tr(Spq) + N |t|^2 + 2 t.(R Sp) - 2 sum_ij R_ij Spq_ji - 2 t.Sq.
ICPX_HD void best_fit(const double H[9], const double mean_tar[3], const double mean_src[3], const double sums[16], double N, double T[16], double* mse_after) {
    double R[9];
    kabsch_rotation(H, R);
    for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = mean_tar[r] - (R[3 * r + 0] * mean_src[0] + R[3 * r + 1] * mean_src[1] + R[3 * r + 2] * mean_src[2]);
    }
    const double* Sp = sums; const double* Sq = sums + 3; const double* Spq = sums + 6;
    double tr = Spq[0] + Spq[4] + Spq[8], tt = 0, tRSp = 0, RS = 0, tSq = 0;
    for (int i = 0; i < 3; i++) {
        tt += T[4 * i + 3] * T[4 * i + 3];
        tSq += T[4 * i + 3] * Sq[i];
        for (int j = 0; j < 3; j++) { tRSp += T[4 * i + 3] * R[3 * i + j] * Sp[j]; RS += R[3 * i + j] * Spq[3 * j + i]; }
    }
    *mse_after = (sums[15] + 2 * tr + N * tt + 2 * tRSp - 2 * RS - 2 * tSq) / N;
}

}  // namespace icpx_solve
