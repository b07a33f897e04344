// The device-wide union-find behind the clustering of the gating graph (tracker.py:961-974): shared by the grow launch's target
// workgroups (mht_fgrow.hip) and by the stateless seam mht_cluster for tables beyond LDS (mht_cluster.hip).
#pragma once
#include "mht_common.h"

namespace mht {

// ---- clustering inside the grow launch (FDyn::uf_epoch, mht_kernels.h) ------------------------------------------------------------
// 64-bit words, all accesses agent-scope atomics (the workgroups of a launch sit on eight XCDs with non-coherent L2s); a word of an
// earlier scan is "empty" (owner) / "no parent" (parent), nothing is cleared between scans.
//   owner[node]  = {epoch, target}: exchanged by atomic max -- whoever finds a word of this scan there shares the node with the target named;
//   parent[t]    = {epoch, ~p}: p < t, a member of t's component.  Linking is ONE returning atomic max per step, no look-ups: propose the
//                  smaller of two targets as parent of the larger; if the larger had a parent already, the smaller of the two candidates
//                  stays and the other one is linked to it next (indices only go down: it ends).  Every member of a component except its
//                  smallest ends up with a parent, so following the parents from any member ends at the smallest member -- the cluster's
//                  label in the reference's order (tracker.py:972-974).
__device__ __forceinline__ void uf_link(unsigned long long* parent, unsigned epoch, int x, int y) {
    int lo = x < y ? x : y, hi = x < y ? y : x;
    while (lo != hi) {
        const unsigned long long old = atomicMax(&parent[hi], ((unsigned long long)epoch << 32) | (unsigned long long)(0xffffffffu - (unsigned)lo));
        if ((unsigned)(old >> 32) != epoch) break;          // hi had no parent
        const int p = (int)(0xffffffffu - (unsigned)old);
        if (p == lo) break;
        hi = p > lo ? p : lo;
        lo = p > lo ? lo : p;
    }
}
// One wavefront, target `pos`: every measurement node of the association set (LDS bitset tb, AW words) exchanges its owner word; the
// targets found there go to an LDS list (conf[0 .. cap), *nconf of them; beyond cap they are linked straight away).
__device__ __forceinline__ void uf_claim(unsigned long long* owner, unsigned long long* parent, unsigned epoch, int pos, const unsigned long long* tb, int AW, int lane,
                                         int* conf, int cap, int* nconf) {
    const unsigned long long mine = ((unsigned long long)epoch << 32) | (unsigned)pos;
    int n = 0;      // (wave-uniform)
    for (int w0 = 0; w0 < AW; w0 += 64) {
        const int w = w0 + lane;
        unsigned long long bits = (w < AW) ? tb[w] : 0ull;
        while (__any(bits != 0ull)) {
            // up to four nodes per lane and round: the atomics go out together, their answers are looked at afterwards
            unsigned long long old[4];
            bool has[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                has[q] = bits != 0ull;
                old[q] = 0ull;
                if (has[q]) {
                    const int b = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    old[q] = atomicMax(&owner[w * 64 + b], mine);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool cf = has[q] && (unsigned)(old[q] >> 32) == epoch && (int)(unsigned)old[q] != pos;
                const unsigned long long bal = __ballot(cf);
                if (cf) {
                    const int i = n + __popcll(bal & ((1ull << lane) - 1ull));
                    if (i < cap) conf[i] = (int)(unsigned)old[q];
                    else uf_link(parent, epoch, pos, (int)(unsigned)old[q]);
                }
                n += __popcll(bal);
            }
        }
    }
    if (lane == 0) *nconf = n < cap ? n : cap;
}


}  // namespace mht
