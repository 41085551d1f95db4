// dlt.h -- 4-point DLT homography: sign-exact null vector of the 8x9 system (utils/outil.py:68-87).
//
// The reference builds A (8x9, float64, entries formed from float32 products) and takes the last row of
// Vh from numpy's LAPACK dgesdd, whose sign decides whether the hypothesis survives the det(H) > 1e-6 gate
// (utils/outil.py:108,113).  For an 8x9 matrix dgesdd runs the unblocked bidiagonalisation dgebd2 (m < n
// branch) and finishes with dormbr('P','R','T'), so that row 9 of Vh is exactly G_1 G_2 ... G_8 e_9, the
// product of dgebd2's right Householder reflectors applied to the last unit vector.
// That is a fixed-length, non-iterative float64 computation: below it is restated with dlarfg / dlarf /
// dlapy2's operation order.  The same header compiles for the host (tests/host_dlt_check.cpp pins it
// against numpy.linalg.svd without a GPU) and for the device (one hypothesis per lane, A in registers).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RFX_HD __host__ __device__ __forceinline__
#else
#define RFX_HD inline
#endif

// det(H) of the 3x3 float32 homography as torch.det evaluates it on the CPU (utils/outil.py:108,113: counts * (det > 1e-6)).
// torch.det = sign * prod(diag(LU)) with LU from LAPACK sgetrf on the row-major storage read as column-major, i.e. of H^T
// (det is transpose-invariant, the rounding is not).  Operation order probed against torch 2.10 + MKL 2024.2 on 3x3 inputs
// (tests/test_oracle.py::test_det3_lu_is_torch_det: bit-equal on every sample): partial pivoting with the first
// maximum; the first column is scaled by the RECIPROCAL of the pivot, the trailing 2x2 block updated with fused
// multiply-adds; the second column divides by the pivot; det = ((u00 * u11) * u22) * sign.  A float32 cofactor expansion of
// the same matrix differs from this by up to ~1.5e-8 when det is near the 1e-6 gate (terms of O(0.1) cancel): 1.4 % of
// the hypotheses within +-10 % of the gate flipped their decision; with the LU order the gate is the oracle's bit for bit.
// Only the fmaf calls below may fuse: the including translation unit is built with -ffp-contract=off.
RFX_HD float rfx_det3_lu_f32(const float* h) {
    float b00 = h[0], b01 = h[3], b02 = h[6];      // B = H^T
    float b10 = h[1], b11 = h[4], b12 = h[7];
    float b20 = h[2], b21 = h[5], b22 = h[8];
    bool neg = false;
    int p = 0;
    float m = fabsf(b00);
    if (fabsf(b10) > m) { p = 1; m = fabsf(b10); }
    if (fabsf(b20) > m) p = 2;
    if (p == 1) { float t; t = b00; b00 = b10; b10 = t; t = b01; b01 = b11; b11 = t; t = b02; b02 = b12; b12 = t; neg = !neg; }
    if (p == 2) { float t; t = b00; b00 = b20; b20 = t; t = b01; b01 = b21; b21 = t; t = b02; b02 = b22; b22 = t; neg = !neg; }
    if (b00 == 0.0f) return 0.0f;                   // singular: a zero on U's diagonal
    const float r = 1.0f / b00;
    const float l1 = b10 * r, l2 = b20 * r;
    b11 = fmaf(-l1, b01, b11); b12 = fmaf(-l1, b02, b12);
    b21 = fmaf(-l2, b01, b21); b22 = fmaf(-l2, b02, b22);
    if (fabsf(b21) > fabsf(b11)) { float t; t = b11; b11 = b21; b21 = t; t = b12; b12 = b22; b22 = t; neg = !neg; }
    if (b11 == 0.0f) return 0.0f;
    const float l = b21 / b11;
    const float u22 = fmaf(-l, b12, b22);
    const float d = (b00 * b11) * u22;
    return neg ? -d : d;
}

// dlapy2: sqrt(x^2 + y^2) without unnecessary overflow
RFX_HD double rfx_dlapy2(double x, double y) {
    const double xa = fabs(x), ya = fabs(y);
    const double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.0) return w;
    const double q = z / w;
    return w * sqrt(1.0 + q * q);
}

// Is the 8x9 system numerically rank deficient?  bd[0..14] = the lower bidiagonal dgebd2 leaves (d1, e1, d2, e2, ..., d8:
// exactly the off-diagonals of its Golub-Kahan form, the 16x16 symmetric tridiagonal with zero diagonal whose eigenvalues are
// +-sigma_i).  One Sturm count at lambda = rel * max|bd|: the matrix has 8 eigenvalues -sigma_i < lambda, so a count above 8
// means sigma_8 < lambda.  Backward stable; 15 divisions.  Why it matters: with sigma_8 / sigma_1 ~ 1e-17 (three matched
// points collinear in BOTH images: common on the cell lattices) the null space is two-dimensional, a reflector is built from
// a vector that is pure rounding noise, and which unit vector of the null space LAPACK returns depends on the summation order
// inside the host BLAS -- it differs between LAPACK builds.  Those hypotheses are flagged so that a caller who needs the
// host's answer bit for bit can re-solve exactly them with the host's LAPACK (rfx_ransac_degenerate_list).
RFX_HD bool rfx_bidiag_rank_deficient(const double bd[15], double rel) {
    double mx = 0.0;
#pragma unroll
    for (int j = 0; j < 15; ++j) mx = fabs(bd[j]) > mx ? fabs(bd[j]) : mx;
    if (mx == 0.0) return true;
    const double lam = rel * mx;
    double q = -lam;
    int neg = 1;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        if (q == 0.0) q = -1e-300;
        q = -lam - bd[j] * bd[j] / q;
        neg += q < 0.0 ? 1 : 0;
    }
    return neg > 8;
}
#define RFX_DLT_RANK_REL 1e-8    /* sigma_8 < 1e-8 * max|bidiagonal|: the null vector is uncertain beyond ~1e-8 */

// src: 4 source points (u', v'), tgt: 4 target points (u, v), float32 as in the reference; h: 9 doubles; bd: NULL or the 15
// bidiagonal entries (d1, e1, ..., d8) for rfx_bidiag_rank_deficient.
RFX_HD void rfx_dlt4_nullvec(const float src[4][2], const float tgt[4][2], double h[9], double* bd = nullptr) {
    double A[8][9];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float u = tgt[p][0], v = tgt[p][1], us = src[p][0], vs = src[p][1];
        // float32 products, widened on store (utils/outil.py:74-81)
        const float vsu = vs * u, vsv = vs * v, usu = -us * u, usv = -us * v;
        double* r0 = A[2 * p];
        double* r1 = A[2 * p + 1];
        r0[0] = 0.0; r0[1] = 0.0; r0[2] = 0.0; r0[3] = -(double)u; r0[4] = -(double)v; r0[5] = -1.0;
        r0[6] = (double)vsu; r0[7] = (double)vsv; r0[8] = (double)vs;
        r1[0] = (double)u; r1[1] = (double)v; r1[2] = 1.0; r1[3] = 0.0; r1[4] = 0.0; r1[5] = 0.0;
        r1[6] = (double)usu; r1[7] = (double)usv; r1[8] = -(double)us;
    }
    double taup[8];
    // dgebd2, m < n: for each i a right reflector G_i (stored in A[i][i+1..8]) then a left reflector H_i
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // ---- dlarfg(9-i, A[i][i], A[i][i+1..8]) ----
        double ss = 0.0;
#pragma unroll
        for (int j = i + 1; j < 9; ++j) ss += A[i][j] * A[i][j];
        const double xnorm = sqrt(ss);
        double tau = 0.0;
        if (xnorm != 0.0) {
            const double alpha = A[i][i];
            const double nrm = rfx_dlapy2(alpha, xnorm);
            const double beta = alpha >= 0.0 ? -nrm : nrm;  // -sign(nrm, alpha), sign(+0) = +
            tau = (beta - alpha) / beta;
            const double sc = 1.0 / (alpha - beta);
#pragma unroll
            for (int j = i + 1; j < 9; ++j) A[i][j] *= sc;
            A[i][i] = beta;
        }
        taup[i] = tau;
        if (i < 7) {
            // ---- dlarf('Right'): rows i+1..7, v = (1, A[i][i+1..8]) on columns i..8 ----
            if (tau != 0.0) {
#pragma unroll
                for (int r = i + 1; r < 8; ++r) {
                    double w = A[r][i];
#pragma unroll
                    for (int j = i + 1; j < 9; ++j) w += A[r][j] * A[i][j];
                    const double tw = tau * w;
                    A[r][i] -= tw;
#pragma unroll
                    for (int j = i + 1; j < 9; ++j) A[r][j] -= tw * A[i][j];
                }
            }
            // ---- dlarfg(7-i, A[i+1][i], A[i+2..7][i]) ----
            double s2 = 0.0;
#pragma unroll
            for (int r = i + 2; r < 8; ++r) s2 += A[r][i] * A[r][i];
            const double ynorm = sqrt(s2);
            if (ynorm != 0.0) {
                const double alpha = A[i + 1][i];
                const double nrm = rfx_dlapy2(alpha, ynorm);
                const double beta = alpha >= 0.0 ? -nrm : nrm;
                const double tauq = (beta - alpha) / beta;
                const double sc = 1.0 / (alpha - beta);
#pragma unroll
                for (int r = i + 2; r < 8; ++r) A[r][i] *= sc;
                A[i + 1][i] = beta;
                // ---- dlarf('Left'): u = (1, A[i+2..7][i]) on rows i+1..7, columns i+1..8 ----
#pragma unroll
                for (int j = i + 1; j < 9; ++j) {
                    double w = A[i + 1][j];
#pragma unroll
                    for (int r = i + 2; r < 8; ++r) w += A[r][i] * A[r][j];
                    const double tw = tauq * w;
                    A[i + 1][j] -= tw;
#pragma unroll
                    for (int r = i + 2; r < 8; ++r) A[r][j] -= tw * A[r][i];
                }
            }
        }
    }
    if (bd) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bd[2 * i] = A[i][i];
            if (i < 7) bd[2 * i + 1] = A[i + 1][i];
        }
    }
    // last row of Vh = G_1 ... G_8 e_9: apply G_8 first
#pragma unroll
    for (int j = 0; j < 9; ++j) h[j] = 0.0;
    h[8] = 1.0;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (taup[i] != 0.0) {
            double w = h[i];
#pragma unroll
            for (int j = i + 1; j < 9; ++j) w += A[i][j] * h[j];
            const double tw = taup[i] * w;
            h[i] -= tw;
#pragma unroll
            for (int j = i + 1; j < 9; ++j) h[j] -= tw * A[i][j];
        }
    }
}
