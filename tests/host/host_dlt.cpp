// Host build of ransac-flow_amd/csrc/dlt.h so that the sign-exact DLT null vector can be pinned against
// numpy.linalg.svd (the reference's utils/outil.py:84) on a GPU-less machine.  Test infrastructure only.
#include "../../ransac-flow_amd/csrc/dlt.h"
extern "C" void rfx_host_dlt4(const float* X, const float* Y, int N, double* h_out, float* H_out) {
    for (int n = 0; n < N; ++n) {
        float src[4][2], tgt[4][2];
        for (int p = 0; p < 4; ++p) {
            src[p][0] = X[(n * 4 + p) * 3 + 0]; src[p][1] = X[(n * 4 + p) * 3 + 1];
            tgt[p][0] = Y[(n * 4 + p) * 3 + 0]; tgt[p][1] = Y[(n * 4 + p) * 3 + 1];
        }
        double h[9];
        rfx_dlt4_nullvec(src, tgt, h);
        for (int j = 0; j < 9; ++j) { h_out[n * 9 + j] = h[j]; H_out[n * 9 + j] = (float)h[j]; }
    }
}

// det(H) as torch.det evaluates it (dlt.h: rfx_det3_lu_f32), for the CPU pin against torch.det.
extern "C" void rfx_host_det3(const float* H, int N, float* out) {
    for (int n = 0; n < N; ++n) out[n] = rfx_det3_lu_f32(H + n * 9);
}

// The rank flag of the DLT kernel (dlt.h: Sturm count on the bidiagonal dgebd2 leaves), for the CPU pin against numpy's
// singular values.
extern "C" void rfx_host_dlt4_rank(const float* X, const float* Y, int N, unsigned char* deficient) {
    for (int n = 0; n < N; ++n) {
        float src[4][2], tgt[4][2];
        for (int p = 0; p < 4; ++p) {
            src[p][0] = X[(n * 4 + p) * 3 + 0]; src[p][1] = X[(n * 4 + p) * 3 + 1];
            tgt[p][0] = Y[(n * 4 + p) * 3 + 0]; tgt[p][1] = Y[(n * 4 + p) * 3 + 1];
        }
        double h[9], bd[15];
        rfx_dlt4_nullvec(src, tgt, h, bd);
        deficient[n] = rfx_bidiag_rank_deficient(bd, RFX_DLT_RANK_REL) ? 1 : 0;
    }
}
