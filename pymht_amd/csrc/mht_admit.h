// Tracker.initiateTarget for a batch of candidates on the device (shared by the forest's own kernels, mht_forest.hip, and the grow
// launch that carries the previous scan's commit AND the admission of what the initiator gave birth to, mht_fgrow.hip).
#pragma once
#include "mht_kernels.h"
#include "mht_commit.h"

namespace mht {

constexpr int ADM_LDS_INTS = 2 + 2048 + 2048;      // LDS scratch of add_targets_body: two scalars, [2048] admitted candidates / leaf offsets of a chunk of targets, [2048] their first nodes

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
    float* ct_Proot;     // constant-turn forest (mht_kernels.h: CtGrow): [Tcap][NP] covariances of the roots born into the newest layer, else null
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
        // every live leaf against every candidate.  The leaf -> node map (target ranges: leaf_off, first) goes through LDS, a chunk of
        // 2047 targets at a time, and every thread has eight leaves in flight: a binary search through global memory per leaf (nine
        // dependent look-ups, then the node's) made this sweep 60 us at the headline size (13 k leaves) -- on every scan with a birth
        int* s_off = sm + 2;            // [2048] leaf_off of the chunk (+ its end)
        int* s_first = sm + 2 + 2048;   // [2047] first node of the chunk's targets
        constexpr int CH = 2047, UN = 8;
        for (int c0 = 0; c0 < nT0; c0 += CH) {
            const int cn = nT0 - c0 < CH ? nT0 - c0 : CH;
            __syncthreads();
            for (int i = tid; i <= cn; i += NT) { s_off[i] = a.tab.leaf_off[c0 + i]; if (i < cn) s_first[i] = a.tab.first[c0 + i]; }
            __syncthreads();
            const int lbeg = s_off[0], lend = s_off[cn];
            for (int i0 = lbeg + tid; i0 < lend; i0 += UN * NT) {
                int nd[UN]; bool ok[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = i0 + u * NT;
                    ok[u] = i < lend;
                    int lo = 0, hi = cn;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
                    nd[u] = ok[u] ? s_first[lo] + (i - s_off[lo]) : 0;
                }
                uint8_t fl[UN]; double lx[UN], ly[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) { fl[u] = a.layer.flags[nd[u]]; lx[u] = a.layer.x[nd[u]]; ly[u] = a.layer.x[(size_t)a.layer.cap + nd[u]]; }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (!ok[u] || (fl[u] & F_DEAD)) continue;      // (F_DEAD: taken out of the tree by similar-state pruning)
                    for (int q = 0; q < an; ++q) {
                        const double dx = lx[u] - a.x0[q * NX], dy = ly[u] - a.x0[q * NX + 1];
                        if (sqrt(dx * dx + dy * dy) < a.thr) a.near[q] = 1;
                    }
                }
            }
        }
        (void)L0;
    }
    __threadfence_block();
    __syncthreads();
    // Sequential admission like the reference (a candidate is also tested against the candidates admitted before it in this
    // call), but the test of one candidate against the admitted ones is spread over the workgroup: a batch of 500 initial
    // targets took 33 ms with one thread walking the O(n^2) pairs.
    int& s_near = sm[0];
    int& s_nadm = sm[1];
    int* s_adm = sm + 2;                        // [2048] candidate indices admitted so far (chunked if more)
    int* s_nearv = sm + 2 + 2048;               // [2048] the candidates' flags of the sweep above (one look-up for all, not one per candidate)
    if (tid == 0) s_nadm = 0;
    for (int q = tid; q < an && q < 2048; q += NT) s_nearv[q] = a.near[q];
    // the forest's counters live in thread 0's registers during the loop (they were a dozen dependent global look-ups per candidate)
    int c_nT = nT0, c_roots = r0, c_L = L0, c_id = 0;
    if (tid == 0) c_id = a.cnt->id_counter;
    __syncthreads();
    for (int q = 0; q < an; ++q) {
        if (tid == 0) s_near = q < 2048 ? s_nearv[q] : a.near[q];
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
            const int ok = !near && c_nT < a.Tcap && c_roots < a.Tcap;
            if (!near && !ok) a.cnt->overflow = 1;
            double xq[NX];
            for (int k = 0; k < NX; ++k) xq[k] = a.x0[q * NX + k];
            const int mq = a.meas[q];
            if (ok) {
                // roots born into a layer live at its end (node root_base + r): the children of a scan are spread over the regions
                // of the node index space below it (fgrow_kernel)
                const int r = c_roots, idx = a.root_base + r, t = c_nT, L = c_L;
                const size_t cap = a.layer.cap;
                const uint8_t fq = a.flags[q];
                for (int k = 0; k < NX; ++k) a.layer.x[k * cap + idx] = xq[k];
                a.layer.cnllr[idx] = 0.0;          // cumulativeNLLR = 0 (pyTarget.py:32)
                a.layer.pd[idx] = a.pd[q];
                a.layer.parent[idx] = -1;
                a.layer.meas[idx] = mq;
                a.layer.cov[idx] = -1;                     // (its key is made below, once the admissions are known)
                a.layer.flags[idx] = fq;
                if (a.mmsi) { a.mmsi[idx] = 0; a.hmmsi[idx] = 0; }
                for (int d = 0; d < a.PD; ++d) { a.path[(size_t)idx * a.PD + d] = -1; a.apath[(size_t)idx * a.PD + d] = -1; }      // (PD = record length here)
                a.tab.id[t] = c_id;
                a.tab.window[t] = a.Nwin;
                a.tab.depth[t] = 0;
                a.tab.shift[t] = 0;
                a.tab.root_scan[t] = a.scan;
                a.tab.root_node[t] = idx;
                a.tab.root_cnllr[t] = 0.0;
                a.tab.root_f32[t] = (fq & F_SCORE_F32) ? 1 : 0;
                a.tab.first[t] = idx;
                a.tab.leaf_off[t] = L;
                a.tab.leaf_off[t + 1] = L + 1;
                if (a.ids) a.ids[q] = c_id;
                c_id += 1; c_roots = r + 1; c_nT = t + 1; c_L = L + 1;
                s_adm[s_nadm & 2047] = q;
                s_nadm += 1;
            } else if (a.ids) {
                a.ids[q] = -1;
            }
            if (a.accepted) a.accepted[q] = (uint8_t)ok;
            if (a.births) {      // the candidate and its fate, for the host mirror (mht_scan_report::births)
                mht_birth_report& b = a.births[q];
                b.id = ok ? c_id - 1 : -1;
                b.meas = mq;
                for (int k = 0; k < NX; ++k) b.x0[k] = xq[k];
                for (int e = 0; e < NP; ++e) b.P0[e] = a.P0[q * NP + e];
            }
        }
        __syncthreads();
    }
    if (tid == 0) { a.cnt->id_counter = c_id; a.cnt->n_roots = c_roots; a.cnt->nT = c_nT; a.cnt->nTv[a.vidx] = c_nT; a.cnt->L = c_L; }
    if (a.hdr && tid == 0) a.hdr->n_births = an;
    // covariance and gains of the admitted roots (what fgrow_kernel's chain workgroups resolve for every other node one scan
    // ahead): the root's covariance by value, and a key of its own -- a pseudo parent id whose miss child is that value
    for (int k = tid; k < s_nadm; k += NT) {
        const int q = s_adm[k & 2047];
        float P[NP];
        for (int e = 0; e < NP; ++e) P[e] = a.P0[q * NP + e];
        if (a.ct_Proot) {      // (nothing is shared by value: the root's covariance goes to its layer's root array, its key names the slot)
            for (int e = 0; e < NP; ++e) a.ct_Proot[(size_t)(r0 + k) * NP + e] = P[e];
            a.layer.cov[a.root_base + r0 + k] = -2 - (r0 + k);
            continue;
        }
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
