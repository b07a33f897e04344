// TEST-ONLY host build of pymht_amd/csrc/mht_math.h (g++ -ffp-contract=off).  It lets the CPU test-suite
// check the kernel arithmetic against the golden vectors bit for bit without a GPU.  Not shipped, not
// used by the product: the product path is the HIP library and fails loudly when that is missing.
#include "../../pymht_amd/csrc/mht_math.h"
#include <cstring>
using namespace mht;

template <typename TS>
static void run(const Model& m, int n, int M, const double* x, const float* P, const float* z, double P_d,
                double* x_bar, float* P_bar, float* P_hat, float* S, float* S_inv, float* K,
                double* nis /*n*M*/, unsigned char* gate /*n*M*/, double* x_hat /*n*M*4*/, double* nllr /*n*M*/) {
    for (int i = 0; i < n; ++i) {
        TS xs[4];
        for (int k = 0; k < 4; ++k) xs[k] = (TS)x[i * 4 + k];
        Predicted<TS> p;
        predict_precalc<TS>(m, xs, P + i * 16, p);
        for (int k = 0; k < 4; ++k) x_bar[i * 4 + k] = (double)p.x_bar[k];
        memcpy(P_bar + i * 16, p.P_bar, 64);
        memcpy(P_hat + i * 16, p.P_hat, 64);
        memcpy(S + i * 4, p.S, 16);
        memcpy(S_inv + i * 4, p.S_inv, 16);
        memcpy(K + i * 8, p.K, 32);
        float lnc = nllr_const(p.S, m.lambda_ex, P_d);
        TS eta2 = (TS)m.eta2;
        for (int j = 0; j < M; ++j) {
            TS zt[2], v;
            bool g = gate_pair<TS>(p.z_hat, p.S_inv, z[j * 2], z[j * 2 + 1], eta2, zt, v);
            nis[(size_t)i * M + j] = (double)v;
            gate[(size_t)i * M + j] = g;
            TS xh[4];
            update_state<TS>(p.x_bar, p.K, zt, xh);
            for (int k = 0; k < 4; ++k) x_hat[((size_t)i * M + j) * 4 + k] = (double)xh[k];
            TS half = (TS)0.5;
            nllr[(size_t)i * M + j] = (double)(half * v + (TS)lnc);
        }
    }
}

extern "C" void mht_host_process(const float* A, const float* Q, const float* C, const float* R, double eta2,
                                 double lambda_ex, int f32state, int n, int M, const double* x, const float* P,
                                 const float* z, double P_d, double* x_bar, float* P_bar, float* P_hat, float* S,
                                 float* S_inv, float* K, double* nis, unsigned char* gate, double* x_hat,
                                 double* nllr) {
    Model m;
    memcpy(m.A, A, 64); memcpy(m.Q, Q, 64); memcpy(m.C, C, 32); memcpy(m.R, R, 16);
    m.eta2 = eta2; m.lambda_ex = lambda_ex;
    if (f32state) run<float>(m, n, M, x, P, z, P_d, x_bar, P_bar, P_hat, S, S_inv, K, nis, gate, x_hat, nllr);
    else run<double>(m, n, M, x, P, z, P_d, x_bar, P_bar, P_hat, S, S_inv, K, nis, gate, x_hat, nllr);
}
