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
        predict_precalc<TS>(m, xs, P + i * 16, p, n == 1);      // (ONE leaf in the call: NumPy's gemv order, like the kernels)
        for (int k = 0; k < 4; ++k) x_bar[i * 4 + k] = (double)p.x_bar[k];
        memcpy(P_bar + i * 16, p.P_bar, 64);
        memcpy(P_hat + i * 16, p.P_hat, 64);
        memcpy(S + i * 4, p.S, 16);
        memcpy(S_inv + i * 4, p.S_inv, 16);
        memcpy(K + i * 8, p.K, 32);
        float lnc = nllr_const(p.S, m.lambda_ex, P_d);
        TS eta2 = (TS)m.eta2;
        int hits = 0;
        for (int j = 0; j < M; ++j) { TS zt[2], v; hits += gate_pair<TS>(p.z_hat, p.S_inv, z[j * 2], z[j * 2 + 1], eta2, zt, v) ? 1 : 0; }
        for (int j = 0; j < M; ++j) {
            TS zt[2], v;
            bool g = gate_pair<TS>(p.z_hat, p.S_inv, z[j * 2], z[j * 2 + 1], eta2, zt, v);
            nis[(size_t)i * M + j] = (double)v;
            gate[(size_t)i * M + j] = g;
            TS xh[4];
            update_state<TS>(p.x_bar, p.K, zt, xh, hits == 1);      // (ONE gated measurement: gemv)
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

// dimension-generic arithmetic (predict_precalc_x), 6 states: tests/golden/g11_kalman6.npz
template <typename TS>
static void run6(const ModelX<6>& m, int n, int M, const double* x, const float* P, const float* z, double P_d,
                 double* x_bar, float* P_bar, float* P_hat, float* S, float* S_inv, float* K,
                 unsigned char* gate /*n*M*/, double* x_hat /*n*M*6*/, double* nllr /*n*M*/) {
    for (int i = 0; i < n; ++i) {
        TS xs[6], xb[6], zh[2];
        for (int k = 0; k < 6; ++k) xs[k] = (TS)x[i * 6 + k];
        float Pb[36], Ph[36], Kk[12], Ss[4], Si[4];
        predict_precalc_x<TS, 6>(m, xs, P + i * 36, xb, zh, Pb, Ph, Kk, Ss, Si, n == 1);
        for (int k = 0; k < 6; ++k) x_bar[i * 6 + k] = (double)xb[k];
        memcpy(P_bar + i * 36, Pb, 144); memcpy(P_hat + i * 36, Ph, 144);
        memcpy(S + i * 4, Ss, 16); memcpy(S_inv + i * 4, Si, 16); memcpy(K + i * 12, Kk, 48);
        const float lnc = nllr_const(Ss, m.lambda_ex, P_d);
        int hits = 0;
        for (int j = 0; j < M; ++j) { TS zt[2], v; hits += gate_pair<TS>(zh, Si, z[j * 2], z[j * 2 + 1], (TS)m.eta2, zt, v) ? 1 : 0; }
        for (int j = 0; j < M; ++j) {
            TS zt[2], v;
            gate[(size_t)i * M + j] = gate_pair<TS>(zh, Si, z[j * 2], z[j * 2 + 1], (TS)m.eta2, zt, v);
            for (int k = 0; k < 6; ++k) x_hat[((size_t)i * M + j) * 6 + k] = (double)update_component_n<TS>(xb[k], Kk[2 * k], Kk[2 * k + 1], zt, hits == 1);
            nllr[(size_t)i * M + j] = (double)((TS)0.5 * v + (TS)lnc);
        }
    }
}
extern "C" void mht_host_process_x6(const float* A, const float* Q, const float* C, const float* R, double eta2, double lambda_ex,
                                    int f32state, int n, int M, const double* x, const float* P, const float* z, double P_d,
                                    double* x_bar, float* P_bar, float* P_hat, float* S, float* S_inv, float* K,
                                    unsigned char* gate, double* x_hat, double* nllr) {
    ModelX<6> m;
    memcpy(m.A, A, 144); memcpy(m.Q, Q, 144); memcpy(m.C, C, 48); memcpy(m.R, R, 16);
    m.eta2 = eta2; m.lambda_ex = lambda_ex;
    if (f32state) run6<float>(m, n, M, x, P, z, P_d, x_bar, P_bar, P_hat, S, S_inv, K, gate, x_hat, nllr);
    else run6<double>(m, n, M, x, P, z, P_d, x_bar, P_bar, P_hat, S, S_inv, K, gate, x_hat, nllr);
}

// np.add.reduce of a 1-D array as similar-state pruning forms the mean score (Sum1D): against NumPy itself in the CPU suite
extern "C" double mht_host_sum1d_f64(const double* v, int n) {
    Sum1D<double> s;
    s.begin(n);
    for (int i = 0; i < n; ++i) s.add(i, v[i]);
    return s.res;
}
extern "C" float mht_host_sum1d_f32(const float* v, int n) {
    Sum1D<float> s;
    s.begin(n);
    for (int i = 0; i < n; ++i) s.add(i, v[i]);
    return s.res;
}

// ---- AIS fusion (csrc/mht_ais_math.h): tests/golden/g19_ais_fusion.npz -------------------------------------------------------
#include "../../pymht_amd/csrc/mht_ais_math.h"
struct HostEmit {
    double* ox; double* oP; int* oradar; double* onllr; int* omsg; int cap; int n;
    void operator()(const double* x, const double* P, int radar, double nllr, int msg) {
        if (n < cap) {
            memcpy(ox + n * 4, x, 32); memcpy(oP + n * 16, P, 128);
            oradar[n] = radar; onllr[n] = nllr; omsg[n] = msg;
        }
        ++n;
    }
};
extern "C" int mht_host_fuse_ais(const float* Cm, const float* R, double eta2, double lambda_ex, const void* groups, int nG, const void* msgs,
                                 int f32state, const double* x, const float* P, double pd, int own, double eta2_ais, double lambda_ais,
                                 const float* z, int M, int cap, double* ox, double* oP, int* oradar, double* onllr, int* omsg) {
    Model m;
    memset(&m, 0, sizeof(m));
    memcpy(m.C, Cm, 32); memcpy(m.R, R, 16);
    m.eta2 = eta2; m.lambda_ex = lambda_ex;
    HostEmit e{ox, oP, oradar, onllr, omsg, cap, 0};
    const AisGroup* g = reinterpret_cast<const AisGroup*>(groups);
    const AisMsg* ms = reinterpret_cast<const AisMsg*>(msgs);
    if (f32state) {
        float xs[4] = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
        return ais_fuse_leaf<float>(m, g, nG, ms, xs, P, pd, own, eta2_ais, lambda_ais, z, M, e);
    }
    return ais_fuse_leaf<double>(m, g, nG, ms, x, P, pd, own, eta2_ais, lambda_ais, z, M, e);
}

// ---- float64 covariance chain of a promoted target (csrc/mht_la64.h): against the reference's kalman.predict / precalc on float64 batches
extern "C" void mht_host_cov_chain64(const float* A, const float* Q, const float* C, const float* R, int n, const double* P,
                                     double* P_bar, double* P_hat, double* S, double* S_inv, double* K) {
    Model m;
    memset(&m, 0, sizeof(m));
    memcpy(m.A, A, 64); memcpy(m.Q, Q, 64); memcpy(m.C, C, 32); memcpy(m.R, R, 16);
    for (int i = 0; i < n; ++i) {
        CovChain64 c;
        cov_chain64(m, P + i * 16, c);
        memcpy(P_bar + i * 16, c.P_bar, 128); memcpy(P_hat + i * 16, c.P_hat, 128);
        memcpy(S + i * 4, c.S, 32); memcpy(S_inv + i * 4, c.S_inv, 32); memcpy(K + i * 8, c.K, 64);
    }
}
extern "C" double mht_host_inv_lapack(int n, const double* s, double* out) {
    return n == 2 ? inv_lapack<2>(s, out) : inv_lapack<4>(s, out);
}
