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

// The piece-tile tables of a tile plan from a table the LOADER made (dmpnn_pack_tiles below): copy, pad the unused slots
// with the (n_edges, n_atoms) sentinel, write the header of a tile plan.  Every entry is checked here for order (a
// violation sets DMPNN_PLAN_NO_PIECE_TILES: the tile kernel then returns NaN; a tile beyond the matrix-pipe limits is
// counted in DMPNN_HDR_NSPILL and takes the tile kernel's generic path); whether a tile
// is CLOSED (holds every edge of its atoms and nothing else) is checked by the tile kernel itself on the batch's own
// index arrays, exactly as for a table from dmpnn_prepare_tiles.
__global__ __launch_bounds__(256) void k_tiles_from_table(const int* __restrict__ tile_row, const int* __restrict__ tile_atom,
                                                          int n_tiles, int nV, int nE, int* __restrict__ plan, PlanLayout L) {
    tiles_from_table_body(tile_row, tile_atom, n_tiles, nV, nE, plan, L);
}

}  // namespace
}  // namespace dmpnn

// HOST function (no device work): greedy packing of consecutive whole molecules into tiles of <= 48 directed edges and
// <= 32 atoms from the two running offsets of a batch.  Writes tile_row / tile_atom [n_tiles + 1] (the last entry is the
// (n_edges, n_atoms) end) and returns n_tiles; a molecule that alone exceeds a tile gets a tile of its own (the tile
// kernel carries it through its generic path); -2 when `cap` entries do not suffice or an argument is bad.
extern "C" int64_t dmpnn_pack_tiles(const int* atom_off, const int* edge_off, int64_t n_mols, int* tile_row, int* tile_atom,
                                    int64_t cap) {
    if (n_mols < 0 || cap < 1 || !tile_row || !tile_atom || (n_mols > 0 && (!atom_off || !edge_off))) return -2;
    int64_t n = 0, m = 0;
    while (m < n_mols) {
        const int a0 = atom_off[m], e0 = edge_off[m];
        int64_t q = m;
        while (q < n_mols && atom_off[q + 1] - a0 <= dmpnn::kMegaBA && edge_off[q + 1] - e0 <= dmpnn::kMegaBM) ++q;
        if (q == m) q = m + 1;                       // molecule m alone exceeds a tile: a tile of its own (generic path)
        if (atom_off[q] == a0 && edge_off[q] == e0) { m = q; continue; }  // only empty molecules: no tile
        if (n + 1 >= cap) return -2;
        tile_row[n] = e0; tile_atom[n] = a0;
        ++n;
        m = q;
    }
    tile_row[n] = n_mols ? edge_off[n_mols] : 0;
    tile_atom[n] = n_mols ? atom_off[n_mols] : 0;
    return n;
}

extern "C" int64_t dmpnn_max_tiles(int64_t n_atoms, int64_t n_edges) { return dmpnn::mega_max_tiles(n_atoms, n_edges); }

extern "C" int dmpnn_prepare_tiles_from_table(const int* tile_row, const int* tile_atom, int64_t n_tiles, int64_t n_atoms,
                                              int64_t n_edges, void* plan, size_t plan_bytes, void* stream) {
    DMPNN_CHECK_ARG(n_atoms >= 0 && n_edges >= 0 && n_tiles >= 0 && n_atoms < (1ll << 31) && n_edges < (1ll << 31), "tiles_from_table: bad sizes");
    DMPNN_CHECK_ARG(plan && (n_tiles == 0 || (tile_row && tile_atom)), "tiles_from_table: NULL pointer");
    const dmpnn::PlanLayout L = dmpnn::plan_layout(n_atoms, n_edges);
    if (plan_bytes < (size_t)L.words * sizeof(int)) {
        dmpnn::set_error("tiles_from_table: plan buffer too small (%zu < %zu bytes)", plan_bytes, (size_t)L.words * sizeof(int));
        return DMPNN_ENOSPC;
    }
    DMPNN_CHECK_ARG(n_tiles <= L.max_mtiles, "tiles_from_table: %lld tiles exceed the launch bound %lld of this batch size",
                    (long long)n_tiles, (long long)L.max_mtiles);
    hipLaunchKernelGGL(dmpnn::k_tiles_from_table, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), tile_row, tile_atom,
                       (int)n_tiles, (int)n_atoms, (int)n_edges, static_cast<int*>(plan), L);
    DMPNN_CHECK_LAUNCH("k_tiles_from_table");
    return DMPNN_OK;
}

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
