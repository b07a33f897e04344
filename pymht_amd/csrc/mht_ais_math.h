// AIS-aided children of a leaf hypothesis: the arithmetic of Tracker.__fuseRadarAndAis (pymht/tracker.py:417-552; file:line
// relative to /root/reference), shared by the HIP kernels (mht_ais.hip) and by tests/hostmath (host build, CPU test-suite only).
//
// Per leaf and per AIS message [x, y, vx, vy] made at time t_a inside the radar period (models/ais.py: C = I4 float64,
// R = sigma^2 I4 float32, sigma = 1 "high accuracy" / 3 otherwise):
//   1. leaf -> t_a:         x1 = Phi(dT1) x,  P1 = Phi(dT1) P Phi(dT1)^T + Q(dT1)             kalman.py:67-70 (tracker.py:449-450)
//   2. AIS precalc:         S1 = P1 + R_ais, S1^-1, K1 = P1 S1^-1, P1^ = P1 - K1 P1           kalman.py:82-101 (:455-459)
//   3. gate the messages:   nis1 = (m - x1)^T S1^-1 (m - x1) <= eta2_ais                     :467-474
//      score                nllr1 = nis1 / 2 + ln(lambda_ais sqrt(det(2 pi S1)))             :479 (P_d = 1)
//   4. update, go on to the scan: xa = x1 + K1 (m - x1);  x2 = Phi(dT2) xa, P2 = Phi(dT2) P1^ Phi(dT2)^T + Q(dT2)   :484-487
//   5. radar precalc + gate against all M radar measurements like any prediction              :488-496
//   6. a child per gated radar measurement, score (nllr1 + nllr2) / 2; none gated: ONE child without a radar measurement,
//      state x2, score nllr1                                                                  :497-526
// dtypes as NumPy promotes them: step 1 in the leaf's OWN dtypes (kalman.predict_single on node.x_0 / node.P_0: float64 state or the
// float32 chain of initiator-born tracks; float32 covariance, or float64 once the target has been promoted); from step 2 on everything
// is float64 (ais.C is float64), INCLUDING the children's covariance, which the forest keeps in float64 (mht_vtab.h: values of two ids).
// Every float64 product is a dgemm / dgemv in OpenBLAS' order (FMA chains, k ascending; gemv: mht_math.h::gemv_row), the two inverses
// are dgesv restated operation by operation (mht_la64.h::inv_lapack): states and covariances of the children are the reference's bit
// for bit (tests/golden/g19_ais_fusion.npz), scores to the NLLR tolerance (numpy's det = exp(sum(log|u_ii|)), its SIMD log).
#pragma once
#include "mht_math.h"
#include "mht_la64.h"

namespace mht {

// One (message time, accuracy class) group of a scan's AIS messages, in the order the reference walks them (tracker.py:447-453):
// the times in the iteration order of the SET of times, high accuracy first; the messages of a group are contiguous.
struct AisGroup {
    float A1[16], Q1[16];      // Phi, Q over dT1 = t_a - (time of the leaves)         models/pv.py:12-24 evaluated by the host
    float A2[16], Q2[16];      // Phi, Q over dT2 = (time of the scan) - t_a
    float r_diag;              // sigma^2 of the group's accuracy class (models/ais.py:9-13)
    int32_t first, count;      // its messages
    int32_t pad;
};
struct AisMsg {
    double state[4];
    int32_t mmsi;              // identity (> 1e8, tracker.py:183-185: distinct within a scan)
    int32_t pad;
};

// matrix x ONE column in float64 (BLAS gemv order, as probed for mht_math.h::gemv_row)
MHT_HD double dgemv4(const double* a, const double* x) {
    const double p0 = a[0] * x[0], p1 = a[1] * x[1], p2 = a[2] * x[2], p3 = a[3] * x[3];
    return (p0 + p2) + (p1 + p3);
}
MHT_HD double dgemv2(const double* a, const double* x) { return fma(a[0], x[0], a[1] * x[1]); }

// steps 1-2 of a (leaf, group): what the gate of step 3 needs
struct AisPre {
    double x1[4];
    double P1[16];     // (float32 values when the leaf's covariance is float32)
    double Sinv[16];
    double lnc1;       // ln(lambda_ais sqrt(det(2 pi S1)))
};
// (a): the prediction at the message's time -- cheap; (b): S^-1 and the score constant -- two thirds of the work of a (leaf, group) pair,
// only done when a message of the group lies inside the gate's bounding box (for a positive definite S: dz_i^2 <= nis * S_ii)
template <typename TS, typename TP>
MHT_HD void ais_pre_a(const AisGroup& g, const TS* x, const TP* P, AisPre& o) {
    TS x1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // A.dot(x): matrix x ONE column (kalman.py:68) -> BLAS gemv order (mht_math.h::gemv_row)
        x1[i] = gemv_row<TS, 4>(g.A1 + i * 4, x);
        o.x1[i] = (double)x1[i];
    }
    TP t[16], at[16], p1[16];      // A.dot(P).dot(A.T) + Q (kalman.py:69) in the covariance's dtype (Phi, Q float32: promoted by NumPy)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) at[i * 4 + j] = (TP)g.A1[j * 4 + i];
    gemm_chain<TP, float, TP, 4, 4, 4>(g.A1, P, t);
    gemm_chain<TP, TP, TP, 4, 4, 4>(t, at, p1);
#pragma unroll
    for (int e = 0; e < 16; ++e) o.P1[e] = (double)(p1[e] + (TP)g.Q1[e]);
}
MHT_HD void ais_pre_b(const AisGroup& g, double lambda_ais, AisPre& o) {
    double S[16];
    const double two_pi = 6.283185307179586;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = o.P1[e] + (((e >> 2) == (e & 3)) ? (double)g.r_diag : 0.0);      // matmul(matmul(C, P), C.T) + R with C = I4: exact
    const double det = inv_lapack<4>(S, o.Sinv);      // det(2 pi S) = (2 pi)^4 det(S): the same elimination serves both (kalman.py:19 scales first: 1e-16 relative)
    o.lnc1 = log((lambda_ais * sqrt((two_pi * two_pi) * (two_pi * two_pi) * det)) / 1.0);
}
template <typename TS, typename TP>
MHT_HD void ais_pre(const AisGroup& g, const TS* x, const TP* P, double lambda_ais, AisPre& o) {
    ais_pre_a<TS, TP>(g, x, P, o);
    ais_pre_b(g, lambda_ais, o);
}

MHT_HD double ais_nis(const AisPre& p, const double* m, double* zt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) zt[i] = m[i] - p.x1[i];
    double nis = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // (zt S^-1)_j zt_j, summed j ascending (kalman.py:25-28)
        double acc = zt[0] * p.Sinv[j];
#pragma unroll
        for (int i = 1; i < 4; ++i) acc = fma(zt[i], p.Sinv[i * 4 + j], acc);
        nis = (j == 0) ? acc * zt[j] : nis + acc * zt[j];
    }
    return nis;
}

// steps 4-5 of a gated (leaf, message): the prediction at the scan's time and its radar gains
struct AisPost {
    double x2[4];
    double P2h[16];     // covariance of every child of this (leaf, message)
    double zhat[2], Sinv[4], K2[8];
    double lnc2;        // ln(lambda_ex sqrt(det(2 pi S2)) / P_d)
};
MHT_HD void ais_post(const AisGroup& g, const Model& mdl, const AisPre& p, const double* zt, double pd, AisPost& o) {
    double K1[16], P1h[16];
    const double* P1 = p.P1;
    gemm_chain<double, double, double, 4, 4, 4>(P1, p.Sinv, K1);            // K = (P C^T) S^-1, C = I
    gemm_chain<double, double, double, 4, 4, 4>(K1, P1, P1h);               // (K C) P
#pragma unroll
    for (int e = 0; e < 16; ++e) P1h[e] = P1[e] - P1h[e];
    double xa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = p.x1[i] + dgemv4(K1 + i * 4, zt);      // tracker.py:484
    double A2[16], A2t[16], t[16], P2[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { A2[i * 4 + j] = (double)g.A2[i * 4 + j]; A2t[j * 4 + i] = (double)g.A2[i * 4 + j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) o.x2[i] = dgemv4(A2 + i * 4, xa);
    gemm_chain<double, double, double, 4, 4, 4>(A2, P1h, t);
    gemm_chain<double, double, double, 4, 4, 4>(t, A2t, P2);
#pragma unroll
    for (int e = 0; e < 16; ++e) P2[e] = P2[e] + (double)g.Q2[e];
    // radar precalc (kalman.py:82-101) in float64; C is the reference's position selector (models/pv.py:9-10): general C via chains
    double C[8], Ct[8], CP[8], S[4], PCt[8], KC[16], KCP[16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { C[i * 4 + j] = (double)mdl.C[i * 4 + j]; Ct[j * 2 + i] = (double)mdl.C[i * 4 + j]; }
    gemm_chain<double, double, double, 2, 4, 4>(C, P2, CP);
    gemm_chain<double, double, double, 2, 4, 2>(CP, Ct, S);
#pragma unroll
    for (int e = 0; e < 4; ++e) S[e] = S[e] + (double)mdl.R[e];
    const double det = inv_lapack<2>(S, o.Sinv);      // np.linalg.inv (kalman.py:91); det(2 pi S) = (2 pi)^2 det(S)
    gemm_chain<double, double, double, 4, 4, 2>(P2, Ct, PCt);
    gemm_chain<double, double, double, 4, 2, 2>(PCt, o.Sinv, o.K2);
    gemm_chain<double, double, double, 4, 2, 4>(o.K2, C, KC);
    gemm_chain<double, double, double, 4, 4, 4>(KC, P2, KCP);
#pragma unroll
    for (int e = 0; e < 16; ++e) o.P2h[e] = P2[e] - KCP[e];
#pragma unroll
    for (int i = 0; i < 2; ++i) o.zhat[i] = dgemv4(C + i * 4, o.x2);
    const double two_pi = 6.283185307179586;
    o.lnc2 = log((mdl.lambda_ex * sqrt(two_pi * two_pi * det)) / pd);
}

// NIS of a radar measurement against a fused prediction, the gate, and the child's state (tracker.py:491-500)
MHT_HD bool ais_radar_gate(const AisPost& p, const Model& mdl, float zx, float zy, double* zt, double& nis) {
    zt[0] = (double)zx - p.zhat[0];
    zt[1] = (double)zy - p.zhat[1];
    const double t0 = fma(zt[1], p.Sinv[2], zt[0] * p.Sinv[0]), t1 = fma(zt[1], p.Sinv[3], zt[0] * p.Sinv[1]);
    nis = t0 * zt[0] + t1 * zt[1];
    return nis <= mdl.eta2;
}
MHT_HD void ais_child_state(const AisPost& p, const double* zt, double* x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = p.x2[i] + dgemv2(p.K2 + i * 2, zt);
}

// All children of one leaf, in the reference's order; `emit(x[4], P[16], radar index or -1, nllr, message index)` is called once
// per child (a counting pass hands in a functor that only counts).  `own` = the identity the track is bound to (0 = none):
// messages of other ships are skipped (pyTarget.py:269-272).  Returns the number of children.
template <typename TS, typename TP, typename EMIT>
MHT_HD int ais_fuse_leaf(const Model& mdl, const AisGroup* groups, int nG, const AisMsg* msgs, const TS* x, const TP* P, double pd,
                         int own, double eta2_ais, double lambda_ais, const float* z, int M, EMIT&& emit) {
    int n = 0;
    for (int gi = 0; gi < nG; ++gi) {
        const AisGroup& g = groups[gi];
        bool any = false;      // (a group none of whose messages can be this track's costs nothing)
        for (int q = 0; q < g.count && !any; ++q) any = (own == 0) || (msgs[g.first + q].mmsi == own);
        if (!any) continue;
        AisPre pre;
        ais_pre_a<TS, TP>(g, x, P, pre);
        // the gate's bounding box in position (a necessary condition, widened by far more than any rounding): most (leaf, group) pairs
        // have no message near them and stop here
        const double bx = sqrt(eta2_ais * (pre.P1[0] + (double)g.r_diag)) * 1.000001 + 1e-9;
        const double by = sqrt(eta2_ais * (pre.P1[5] + (double)g.r_diag)) * 1.000001 + 1e-9;
        any = false;
        for (int q = 0; q < g.count && !any; ++q) {
            const AisMsg& m = msgs[g.first + q];
            any = (own == 0 || m.mmsi == own) && fabs(m.state[0] - pre.x1[0]) <= bx && fabs(m.state[1] - pre.x1[1]) <= by;
        }
        if (!any) continue;
        ais_pre_b(g, lambda_ais, pre);
        for (int q = 0; q < g.count; ++q) {
            const AisMsg& m = msgs[g.first + q];
            if (own != 0 && m.mmsi != own) continue;
            if (!(fabs(m.state[0] - pre.x1[0]) <= bx && fabs(m.state[1] - pre.x1[1]) <= by)) continue;
            double zt[4];
            const double nis1 = ais_nis(pre, m.state, zt);
            if (!(nis1 <= eta2_ais)) continue;
            const double nllr1 = 0.5 * nis1 + pre.lnc1;
            AisPost post;
            ais_post(g, mdl, pre, zt, pd, post);
            int hits = 0;
            for (int j = 0; j < M; ++j) {
                double zr[2], nis2;
                if (!ais_radar_gate(post, mdl, z[2 * j], z[2 * j + 1], zr, nis2)) continue;
                double xc[4];
                ais_child_state(post, zr, xc);
                const double nllr2 = 0.5 * nis2 + post.lnc2;
                emit(xc, post.P2h, j, 0.5 * nllr1 + 0.5 * nllr2, g.first + q);
                ++hits;
            }
            if (hits == 0) { emit(post.x2, post.P2h, -1, nllr1, g.first + q); hits = 1; }
            n += hits;
        }
    }
    return n;
}

}  // namespace mht
