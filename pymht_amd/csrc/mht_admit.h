// Tracker.initiateTarget for a batch of candidates on the device (shared by the forest's own kernels, mht_forest.hip, and the grow
// launch that carries the previous scan's commit AND the admission of what the initiator gave birth to, mht_fgrow.hip).
#pragma once
#include "mht_kernels.h"
#include "mht_commit.h"

namespace mht {

constexpr int ADM_LDS_INTS = 2 + 2048;      // LDS scratch of add_targets_body

// Tracker.initiateTarget (tracker.py:147-160) for a batch of candidates, sequentially like the reference
struct AddArgs {
    int n; const double* x0; const float* P0; const uint8_t* flags; const double* pd; const int32_t* meas;
    int check; double thr;
    mht_nodes layer;     // newest layer
    TTable tab; int32_t* path; int32_t* apath; int PD;
    FCounts* cnt; int scan; int Nwin; int Tcap;
    int vidx;            // version index of `tab` (FCounts::nTv)
    uint8_t* accepted; int32_t* ids; int32_t* near;   // near: [n] scratch
    Model model; VTab vt; int root_base;
    const int32_t* n_dev;      // number of candidates in device memory (or null: n)
    int32_t* mmsi; int32_t* hmmsi;      // AIS forest: identities of the newest layer's nodes (a root has none), else null
    ReportHeader* hdr; mht_birth_report* births;      // report block of the device initiator's candidates (or null)   // gains of the new roots (row cov_base + r of the newest layer's gain table); first root node
};

// Tracker.initiateTarget (tracker.py:147-160) for a batch of candidates.  The test against the existing leaves
// (pyTarget.haveNoNeightbours, pyTarget.py:181-189) runs for all candidates in one parallel sweep; the candidates are
// then admitted sequentially, each also tested against the ones admitted before it, like the reference's loop.
template <int NT>
__device__ __forceinline__ void add_targets_body(const AddArgs& a, int* sm) {
    const int tid = threadIdx.x;
    const int nT0 = a.cnt->nT, L0 = a.cnt->L, r0 = a.cnt->n_roots;
    int an = a.n;
    if (a.n_dev) { const int nd = *a.n_dev; an = nd < an ? nd : an; }
    if (an <= 0) {          // nothing to admit (the usual case behind the device initiator)
        if (a.hdr && tid == 0) a.hdr->n_births = 0;
        return;
    }
    for (int q = tid; q < an; q += NT) a.near[q] = 0;
    __syncthreads();
    if (a.check) {
        for (int i = tid; i < L0; i += NT) {
            // leaf i -> node: linear scan over targets is avoided by walking the ranges: (first, leaf_off) lookup
            int lo = 0, hi = nT0;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.tab.leaf_off[mid] <= i) lo = mid; else hi = mid; }
            const int nd = a.tab.first[lo] + (i - a.tab.leaf_off[lo]);
            if (a.layer.flags[nd] & F_DEAD) continue;      // (taken out of the tree by similar-state pruning)
            const double lx = a.layer.x[nd], ly = a.layer.x[(size_t)a.layer.cap + nd];
            for (int q = 0; q < an; ++q) {
                const double dx = lx - a.x0[q * NX], dy = ly - a.x0[q * NX + 1];
                if (sqrt(dx * dx + dy * dy) < a.thr) a.near[q] = 1;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // Sequential admission like the reference (a candidate is also tested against the candidates admitted before it in this
    // call), but the test of one candidate against the admitted ones is spread over the workgroup: a batch of 500 initial
    // targets took 33 ms with one thread walking the O(n^2) pairs.
    int& s_near = sm[0];
    int& s_nadm = sm[1];
    int* s_adm = sm + 2;                        // [2048] candidate indices admitted so far (chunked if more)
    if (tid == 0) s_nadm = 0;
    __syncthreads();
    for (int q = 0; q < an; ++q) {
        if (tid == 0) s_near = a.near[q];
        __syncthreads();
        if (a.check && !s_near) {
            const double qx = a.x0[q * NX], qy = a.x0[q * NX + 1];
            const int na = s_nadm;
            int hit = 0;
            for (int i = tid; i < na; i += NT) {
                const int pc = s_adm[i & 2047];
                const double dx = a.x0[pc * NX] - qx, dy = a.x0[pc * NX + 1] - qy;
                if (sqrt(dx * dx + dy * dy) < a.thr) hit = 1;
            }
            if (hit) s_near = 1;
        }
        __syncthreads();
        if (tid == 0) {   // (admission is sequential like the reference's loop)
            const int near = s_near;
            const int ok = !near && a.cnt->nT < a.Tcap && a.cnt->n_roots < a.Tcap;
            if (!near && !ok) a.cnt->overflow = 1;
            if (ok) {
                // roots born into a layer live at its end (node root_base + r): the children of a scan are spread over the regions
                // of the node index space below it (fgrow_kernel)
                const int r = a.cnt->n_roots, idx = a.root_base + r, t = a.cnt->nT, L = a.cnt->L;
                const size_t cap = a.layer.cap;
                for (int k = 0; k < NX; ++k) a.layer.x[k * cap + idx] = a.x0[q * NX + k];
                a.layer.cnllr[idx] = 0.0;          // cumulativeNLLR = 0 (pyTarget.py:32)
                a.layer.pd[idx] = a.pd[q];
                a.layer.parent[idx] = -1;
                a.layer.meas[idx] = a.meas[q];
                a.layer.cov[idx] = -1;                     // (its key is made below, once the admissions are known)
                a.layer.flags[idx] = a.flags[q];
                if (a.mmsi) { a.mmsi[idx] = 0; a.hmmsi[idx] = 0; }
                for (int d = 0; d < a.PD; ++d) { a.path[(size_t)idx * a.PD + d] = -1; a.apath[(size_t)idx * a.PD + d] = -1; }      // (PD = record length here)
                a.tab.id[t] = a.cnt->id_counter;
                a.tab.window[t] = a.Nwin;
                a.tab.depth[t] = 0;
                a.tab.shift[t] = 0;
                a.tab.root_scan[t] = a.scan;
                a.tab.root_node[t] = idx;
                a.tab.root_cnllr[t] = 0.0;
                a.tab.root_f32[t] = (a.flags[q] & F_SCORE_F32) ? 1 : 0;
                a.tab.first[t] = idx;
                a.tab.leaf_off[t] = L;
                a.tab.leaf_off[t + 1] = L + 1;
                if (a.ids) a.ids[q] = a.cnt->id_counter;
                a.cnt->id_counter += 1;
                a.cnt->n_roots = r + 1;
                a.cnt->nT = t + 1;
                a.cnt->nTv[a.vidx] = t + 1;
                a.cnt->L = L + 1;
                s_adm[s_nadm & 2047] = q;
                s_nadm += 1;
            } else if (a.ids) {
                a.ids[q] = -1;
            }
            if (a.accepted) a.accepted[q] = (uint8_t)ok;
            if (a.births) {      // the candidate and its fate, for the host mirror (mht_scan_report::births)
                mht_birth_report& b = a.births[q];
                b.id = ok ? a.cnt->id_counter - 1 : -1;
                b.meas = a.meas[q];
                for (int k = 0; k < NX; ++k) b.x0[k] = a.x0[q * NX + k];
                for (int e = 0; e < NP; ++e) b.P0[e] = a.P0[q * NP + e];
            }
        }
        __syncthreads();
    }
    if (a.hdr && tid == 0) a.hdr->n_births = an;
    // covariance and gains of the admitted roots (what fgrow_kernel's chain workgroups resolve for every other node one scan
    // ahead): the root's covariance by value, and a key of its own -- a pseudo parent id whose miss child is that value
    for (int k = tid; k < s_nadm; k += NT) {
        const int q = s_adm[k & 2047];
        float P[NP];
        for (int e = 0; e < NP; ++e) P[e] = a.P0[q * NP + e];
        const int id0 = vt_find_or_insert(a.vt, P, a.pd[q]);
        const unsigned pid = atomicAdd(a.vt.count, 1u);
        if (pid >= (unsigned)a.vt.vcap) { *a.vt.overflow = 1; continue; }
        const int key = 2 * (int)pid;
        float4 rec[GKQ];
        vt_gains(a.model, P, a.pd[q], rec);
        for (int e = 0; e < GKQ; ++e) a.vt.Gk[(size_t)key * GKQ + e] = rec[e];
        a.vt.child[key] = id0;
        a.layer.cov[a.root_base + r0 + k] = key;      // (admissions are sequential: the k-th took root r0 + k)
    }
}

}  // namespace mht
