// f3 — batching on the device: the index half of BatchMolGraph.__post_init__ (chemprop/data/collate.py:37-62).
//
// Reference: per molecule i, `edge_index_i + num_nodes`, `rev_edge_index_i + num_edges`, `[i] * n_atoms_i`, each a
// numpy temporary, then np.hstack / np.concatenate, torch.from_numpy(...).long() and FIVE host-to-device copies
// (collate.py:68-73) of which three carry int64 indices.
//
// MI355X form: the loader packs a batch into ONE pinned buffer (chemprop_amd/data.py: molecule-LOCAL int32 indices,
// the two running offsets, V, E), one copy brings it over, and this kernel writes the three int64 index tensors the
// block's boundary expects.  V and E are used in place (views of the copied buffer).  HBM bound: 12 B read and 24 B
// written per directed edge, 8 B written per atom — a few hundred nanoseconds at QM9-512; integer work, bit-exact.
//
//   edge e of molecule m (edge_off[m] <= e < edge_off[m+1], found by bisection over the n_mols + 1 offsets):
//       edge_index[0][e] = src[e] + atom_off[m]     edge_index[1][e] = dst[e] + atom_off[m]
//       rev_edge_index[e] = rev[e] + edge_off[m]
//   atom a:  batch[a] = m with atom_off[m] <= a < atom_off[m+1]
//
// A local id outside its molecule is NOT clamped: it comes out as an id outside the molecule's range or of another
// molecule, which the plan kernels' validation flags (DMPNN_PLAN_RANGE_ERROR / ASYMMETRIC) exactly as for a
// hand-built batch.  Molecules without atoms or bonds are fine (equal consecutive offsets).
#include "dmpnn_common.hpp"

namespace dmpnn {
namespace {

// last m in [0, n_mols) with off[m] <= i   (off is non-decreasing, off[0] == 0, i < off[n_mols])
__device__ __forceinline__ int owner(const int* __restrict__ off, int n_mols, int i) {
    int lo = 0, hi = n_mols;  // invariant: off[lo] <= i < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_collate(const int* __restrict__ atom_off, const int* __restrict__ edge_off, int n_mols,
                                                 const int* __restrict__ src, const int* __restrict__ dst,
                                                 const int* __restrict__ rev, int nV, int nE, long long* __restrict__ edge_index,
                                                 long long* __restrict__ rev_edge_index, long long* __restrict__ batch) {
    const int n = nE > nV ? nE : nV;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (i < nE) {
            const int s = src[i], d = dst[i], r = rev[i];  // (issued before the dependent bisection loads)
            const int m = owner(edge_off, n_mols, i);
            const long long a0 = atom_off[m], e0 = edge_off[m];
            edge_index[i] = s + a0;
            edge_index[(long long)nE + i] = d + a0;
            rev_edge_index[i] = r + e0;
        }
        if (i < nV) batch[i] = owner(atom_off, n_mols, i);
    }
}

}  // namespace
}  // namespace dmpnn

extern "C" int dmpnn_collate(const int* atom_off, const int* edge_off, int64_t n_mols, const int* src, const int* dst,
                             const int* rev, int64_t n_atoms, int64_t n_edges, int64_t* edge_index, int64_t* rev_edge_index,
                             int64_t* batch, void* stream) {
    DMPNN_CHECK_ARG(n_mols >= 0 && n_atoms >= 0 && n_edges >= 0 && n_atoms < (1ll << 31) && n_edges < (1ll << 31) && n_mols < (1ll << 31),
                    "collate: bad sizes");
    if (n_atoms == 0 && n_edges == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(n_mols > 0 && atom_off && edge_off, "collate: atoms / edges without molecules");
    DMPNN_CHECK_ARG(n_edges == 0 || (src && dst && rev && edge_index && rev_edge_index), "collate: NULL edge arrays");
    DMPNN_CHECK_ARG(n_atoms == 0 || batch, "collate: NULL batch");
    const int64_t n = n_edges > n_atoms ? n_edges : n_atoms;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dmpnn::k_collate, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), atom_off, edge_off,
                       (int)n_mols, src, dst, rev, (int)n_atoms, (int)n_edges, reinterpret_cast<long long*>(edge_index),
                       reinterpret_cast<long long*>(rev_edge_index), reinterpret_cast<long long*>(batch));
    DMPNN_CHECK_LAUNCH("k_collate");
    return DMPNN_OK;
}
