// K0, tile plan of a batch BEYOND the single-workgroup plan: the piece-tile tables of the whole-forward tile kernel from
// the batch vector (BatchMolGraph.batch, data/collate.py:52,62: the molecule id of every atom, non-decreasing), by three
// small multi-workgroup launches instead of one workgroup's LDS:
//
//   k_large_bounds   one thread per atom / edge: first atom aoff[m] and first directed edge eoff[m] of every molecule from the
//                    boundaries of batch[.] and batch[dst[.]]  (edges of a molecule are contiguous and in molecule order,
//                    collate.py:51-56), gaps (molecules without edges) filled by the boundary thread;
//   k_large_blocks   one wave per block of 64 consecutive molecules: lane l holds "the furthest molecule that still fits a
//                    tile started at molecule base + l" (<= 48 directed edges, <= 32 atoms), the chain of tile starts of
//                    the block is walked with v_readlane hops (a tile starts at every block start: at most one
//                    under-filled tile per block), the block's tile starts go to a scratch list, their number to bcnt;
//   k_large_finish   one wave per block again: rank of the block = sum of the counts before it, its tiles copied to
//                    their place in mtile_row / mtile_atom; every wave also pads its share of the unused slots with the
//                    (n_edges, n_atoms) sentinel; the last writes the tile-plan header.
//
// The same tables as the single-workgroup kernels produce for small batches (blocked greedy packing; restated in
// oracle/collate_numpy.py: blocked_molecule_tiles).  Scratch lives in the arrays of the plan a tile plan never fills
// (src ... ident).  A molecule larger than a tile becomes a tile of its own (counted in DMPNN_HDR_NSPILL; the tile kernel
// runs its generic path on it); more tiles than the launch bound, or a batch vector that is not
// 0 .. n_mols-1 non-decreasing gives DMPNN_PLAN_NO_PIECE_TILES (the tile kernel then returns NaN); everything else a
// wrong table could do is caught by the tile kernel's own closure check on the batch's index arrays.
#include "dmpnn_common.hpp"

namespace dmpnn {
namespace {

struct LargeScratch {
    int64_t aoff, eoff, bcnt, brow, batom, end;  // word offsets inside the plan
    int nblk_max;
};

LargeScratch large_scratch(const PlanLayout& L, int64_t nV) {
    LargeScratch S;
    S.nblk_max = (int)(nV / 64 + 1);
    int64_t o = L.src;
    S.aoff = o; o += align4(nV + 2);
    S.eoff = o; o += align4(nV + 2);
    S.bcnt = o; o += align4(S.nblk_max + 1);
    S.brow = o; o += (int64_t)S.nblk_max * 64;
    S.batom = o; o += (int64_t)S.nblk_max * 64;
    S.end = o;
    return S;
}

constexpr int kNonMono = 1 << 30;   // bcnt bit: the offsets of the block are not monotone (not a batch vector)
constexpr int kCountMask = 0xFF;    // bcnt bits 0..7: tiles of the block (<= 64); bits 8..15: of which oversize (spill) tiles

// number of molecules from the last entry of the batch vector, clamped so that every kernel stays inside its arrays
__device__ __forceinline__ int mol_count(const long long* __restrict__ batch, int nV) {
    const long long last = batch[nV - 1];
    return last < 0 ? 0 : (last >= nV ? nV : (int)last + 1);
}

// One thread per atom and per edge instead of two binary searches per molecule (2 x 18 DEPENDENT L2 reads: 25 us at 4 096
// molecules, as long as the two kernels behind it together): an atom whose molecule differs from its predecessor's IS that
// molecule's first atom; the same for an edge through the molecule of its destination atom.  Molecules without atoms / edges in
// between (a single atom has no edge) take the offset of the next boundary: the boundary thread fills the gap.  For a batch
// vector that is not non-decreasing some entries stay unwritten or contradict each other — k_large_blocks flags offsets that are
// not monotone, and whatever table comes out of the rest is checked by the tile kernel itself (closure, ranges).
__global__ __launch_bounds__(256) void k_large_bounds(const long long* __restrict__ batch, const long long* __restrict__ dst, int nV,
                                                      int nE, int* __restrict__ plan, LargeScratch S) {
    const int n_mols = mol_count(batch, nV);
    int* aoff = plan + S.aoff;
    int* eoff = plan + S.eoff;
    auto mol_of_atom = [&](int v) -> long long {
        const long long m = batch[v];
        return m < 0 ? 0 : (m >= n_mols ? n_mols - 1 : m);
    };
    auto mol_of_edge = [&](int e) -> long long {
        long long d = dst[e];
        d = d < 0 ? 0 : (d >= nV ? nV - 1 : d);  // (an id out of range: the tile kernel flags the edge itself)
        return mol_of_atom((int)d);
    };
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= (nV > nE ? nV : nE); i += stride) {
        if (i < nV) {
            const long long m = mol_of_atom(i), pm = i > 0 ? mol_of_atom(i - 1) : -1;
            for (long long mm = pm + 1; mm <= m; ++mm) aoff[mm] = i;   // (empty unless i is a boundary; one iteration for a dense batch vector)
        } else if (i == nV) {
            aoff[n_mols] = nV;
        }
        if (i < nE) {
            const long long m = mol_of_edge(i), pm = i > 0 ? mol_of_edge(i - 1) : -1;
            for (long long mm = pm + 1; mm <= m; ++mm) eoff[mm] = i;
        } else if (i == nE) {
            const long long pm = nE > 0 ? mol_of_edge(nE - 1) : -1;
            for (long long mm = pm + 1; mm <= n_mols; ++mm) eoff[mm] = nE;   // molecules behind the last edge (and the end marker)
        }
    }
}

__global__ __launch_bounds__(256) void k_large_blocks(const long long* __restrict__ batch, int nV, int* __restrict__ plan, LargeScratch S) {
    const int n_mols = mol_count(batch, nV);
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: the chain cursor and the mask below in scalar registers — dmpnn_prepare.hip, pack_pieces)
    const int base = blk * 64;
    if (base >= n_mols) return;
    const int* aoff = plan + S.aoff;
    const int* eoff = plan + S.eoff;
    const int p = base + lane;
    const bool valid = p < n_mols;
    int nx = n_mols, a0 = 0, e0 = 0, over = 0, bad = 0;
    if (valid) {
        a0 = aoff[p]; e0 = eoff[p];
        int q = p + 1;  // molecules p .. q-1 fit one tile
        if (aoff[q] < a0 || eoff[q] < e0) {
            bad = 1;    // the offsets are not monotone: not a batch vector
        } else if (aoff[q] - a0 > kMegaBA || eoff[q] - e0 > kMegaBM) {
            over = 1;   // molecule p alone exceeds a tile: a tile of its own (the tile kernel's generic path)
        } else {
            while (q < n_mols && aoff[q + 1] - a0 <= kMegaBA && eoff[q + 1] - e0 <= kMegaBM && aoff[q + 1] >= aoff[q] && eoff[q + 1] >= eoff[q]) ++q;
        }
        nx = q;
    }
    const int lim = base + 64 < n_mols ? base + 64 : n_mols;
    unsigned long long mask = 0ull;
    int es = base;  // wave-uniform chain cursor
    while (es < lim) {
        mask |= 1ull << (es - base);
        es = __builtin_amdgcn_readlane(nx, es - base);
    }
    const bool on = (mask >> lane) & 1ull;
    const int r = __popcll(mask & ((1ull << lane) - 1ull));
    if (on) {
        plan[S.brow + (int64_t)blk * 64 + r] = e0;
        plan[S.batom + (int64_t)blk * 64 + r] = a0;
    }
    const unsigned long long overs = __ballot(over != 0), any_bad = __ballot(bad != 0);
    if (lane == 0) plan[S.bcnt + blk] = __popcll(mask) | (__popcll(overs & mask) << 8) | (any_bad ? kNonMono : 0);
}

__global__ __launch_bounds__(256) void k_large_finish(const long long* __restrict__ batch, int nV, int nE, int* __restrict__ plan,
                                                      PlanLayout L, LargeScratch S) {
    const long long last = batch[nV - 1];
    const int n_mols = mol_count(batch, nV);
    const bool bad_batch = last < 0 || last >= nV;  // (molecule ids are 0 .. n_mols-1 <= n_atoms-1)
    const int nblk = (n_mols + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const int* bcnt = plan + S.bcnt;
    int* mrow = plan + L.mtile_row;
    int* matom = plan + L.mtile_atom;
    const int slots = (int)L.max_mtiles + 2;
    // one pass over the block counts: rank of this wave's block, the total, the oversize bits
    int rank = 0, total = 0, nonmono = 0, n_spill = 0;
    for (int b2 = lane; b2 < nblk; b2 += 64) {
        const int c = bcnt[b2];
        nonmono |= c & kNonMono;
        const int n = c & kCountMask;
        n_spill += (c >> 8) & kCountMask;
        total += n;
        if (b2 < wave) rank += n;
    }
    for (int off = 32; off > 0; off >>= 1) {
        rank += __shfl_xor(rank, off);
        total += __shfl_xor(total, off);
        n_spill += __shfl_xor(n_spill, off);
        nonmono |= __shfl_xor(nonmono, off);
    }
    const bool bad = bad_batch || nonmono != 0 || total > (int)L.max_mtiles || total == 0;
    if (!bad && wave < nblk) {
        const int n = bcnt[wave] & kCountMask;
        if (lane < n) {
            mrow[rank + lane] = plan[S.brow + (int64_t)wave * 64 + lane];
            matom[rank + lane] = plan[S.batom + (int64_t)wave * 64 + lane];
        }
    }
    // the unused slots (all of them when the plan is bad): the (n_edges, n_atoms) sentinel
    const int first = bad ? 0 : total;
    for (int t = first + wave * 64 + lane; t < slots; t += n_waves * 64) { mrow[t] = nE; matom[t] = nV; }
    if (wave == 0 && lane < DMPNN_HDR_WORDS) {
        int v = 0;
        if (lane == DMPNN_HDR_FLAGS) v = (bad ? PLAN_NO_PIECE_TILES : 0) | PLAN_TILES_ONLY;
        if (lane == DMPNN_HDR_NMTILES) v = bad ? 0 : total;
        if (lane == DMPNN_HDR_NSPILL) v = bad ? 0 : n_spill;
        if (lane == DMPNN_HDR_LIGHT) v = 2;
        if (lane == DMPNN_HDR_NATOMS) v = nV;
        if (lane == DMPNN_HDR_NEDGES) v = nE;
        if (lane == DMPNN_HDR_TILE_STRIDE) v = kFusedBM;
        plan[lane] = v;
    }
}

}  // namespace

bool tiles_large_fits(int64_t nV, int64_t nE) {
    if (nV <= 0 || nV >= (1ll << 30) || nE >= (1ll << 30)) return false;
    const PlanLayout L = plan_layout(nV, nE);
    return large_scratch(L, nV).end <= L.tile_row;  // the scratch fits the arrays a tile plan leaves unused
}

int launch_prepare_tiles_large(const int64_t* edge_index, const int64_t* batch, int64_t nV64, int64_t nE64, int* plan, hipStream_t s) {
    const int nV = (int)nV64, nE = (int)nE64;
    const PlanLayout L = plan_layout(nV, nE);
    const LargeScratch S = large_scratch(L, nV);
    const long long* b = reinterpret_cast<const long long*>(batch);
    const long long* dst = reinterpret_cast<const long long*>(edge_index) + nE;
    int blocks = ((nV > nE ? nV : nE) + 1 + 255) / 256;  // (one thread per atom / edge)
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_large_bounds, dim3((unsigned)blocks), dim3(256), 0, s, b, dst, nV, nE, plan, S);
    DMPNN_CHECK_LAUNCH("k_large_bounds");
    const unsigned wgs = (unsigned)((S.nblk_max + 3) / 4);
    hipLaunchKernelGGL(k_large_blocks, dim3(wgs), dim3(256), 0, s, b, nV, plan, S);
    DMPNN_CHECK_LAUNCH("k_large_blocks");
    hipLaunchKernelGGL(k_large_finish, dim3(wgs), dim3(256), 0, s, b, nV, nE, plan, L, S);
    DMPNN_CHECK_LAUNCH("k_large_finish");
    return DMPNN_OK;
}

}  // namespace dmpnn
