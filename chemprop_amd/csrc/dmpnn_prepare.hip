// K0 — graph plan for one BatchMolGraph (chemprop/data/collate.py:13-73 is the input contract).
//
// The reference has no counterpart: it re-materialises a dense [E, h] int64 index for every
// scatter (mixins.py:12, base.py:208).  Here the int64 COO arrays are narrowed to int32 once per
// batch and a STABLE incoming-edge CSR keyed by destination atom is built, so every later kernel
// walks an atom's incoming rows in increasing edge id — the order of the reference's sequential
// scatter_reduce_ — with coalesced row reads and no atomics on the data path.  The plan also
// carries the same graph in CSR-ROW coordinates (row i = edge perm[i]) and a table of row tiles
// that hold whole destination atoms: the fused forward keeps its edge tensors in row order and
// forms the segment sums in the epilogue of the contraction that produced them.
//
// All integer work, latency bound.  Two builds of the same algorithm:
//   * k_prepare_small: ONE workgroup, every intermediate (dst, rev, perm, inv as uint16, the
//     per-atom counters) in LDS — a 512-molecule QM9 batch (E ~ 9.6k, V ~ 4.6k) needs ~110 KB;
//     global memory sees one batched read of the int64 arrays and one pass of int32 writes.
//   * the general path: count | scan | fill | sort | inverse | rows+tiles, seven short launches.
#include <atomic>
#include <limits.h>

#include "dmpnn_common.hpp"
#include "dmpnn_mega16_impl.hpp"   // (mega16::SplitArgs / split_weights_wave: the weight pre-split rides in K0's launch)

namespace dmpnn {

extern thread_local long long* g_debug_stamps;

namespace {

constexpr int kBlock = 256;

// keep_mtiles: the molecule tiles are already in the plan (launch_prepare_tiles_large ran before on the same stream):
// their three header words survive — the verdict bit of the flag word, the tile count, the oversize count
__global__ void k_plan_init(int* __restrict__ plan, int64_t cursor_off, int nV, int nE, int keep_mtiles) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (tid < DMPNN_HDR_WORDS) {
        int v = 0;
        if (tid == DMPNN_HDR_NATOMS) v = nV;
        if (tid == DMPNN_HDR_NEDGES) v = nE;
        if (keep_mtiles) {
            if (tid == DMPNN_HDR_FLAGS) v = plan[tid] & PLAN_NO_PIECE_TILES;
            if (tid == DMPNN_HDR_NMTILES || tid == DMPNN_HDR_NSPILL) v = plan[tid];
        }
        plan[tid] = v;
    }
    for (int64_t i = tid; i < nV; i += (int64_t)gridDim.x * blockDim.x) plan[cursor_off + i] = 0;
}

// One thread per directed edge: narrow to int32, range-check, validate the symmetric-graph
// invariants (rev involution, src(rev e) == dst(e), dst(rev e) == src(e)) and histogram dst.
__global__ void k_convert_count(const int64_t* __restrict__ edge_index,
                                const int64_t* __restrict__ rev64, int* __restrict__ plan,
                                PlanLayout L, int nV, int nE) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    if (e < nE) {
        int64_t s = edge_index[e], d = edge_index[(int64_t)nE + e], r = rev64[e];
        if (s < 0 || s >= nV || d < 0 || d >= nV || r < 0 || r >= nE) {
            bad = PLAN_RANGE_ERROR | PLAN_ASYMMETRIC;
            s = s < 0 ? 0 : (s >= nV ? nV - 1 : s);
            d = d < 0 ? 0 : (d >= nV ? nV - 1 : d);
            r = r < 0 ? 0 : (r >= nE ? nE - 1 : r);
        } else {
            const int64_t rr = rev64[r];
            const int64_t sr = edge_index[r], dr = edge_index[(int64_t)nE + r];
            if (rr != e || sr != d || dr != s) bad = PLAN_ASYMMETRIC;
        }
        plan[L.src + e] = (int)s;
        plan[L.dst + e] = (int)d;
        plan[L.rev + e] = (int)r;
        atomicAdd(&plan[L.cursor + d], 1);
    }
    // one atomicOr per wave at most
    const unsigned long long any = __ballot(bad != 0);
    if (any) {
        int v = bad;
        for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) atomicOr(&plan[DMPNN_HDR_FLAGS], v);
    }
}

// Exclusive scan of the per-atom in-degree -> row_ptr (and the fill cursor), two launches: per-block totals (8192 atoms per
// block), then every block rebuilds its prefix on top of the totals before it.  (One workgroup walking the whole array took
// 220 us at 166 k atoms: 10 % of a large-batch forward.)  `part`: scratch of one int per block (the piece-tile table,
// rewritten later by k_rows_tiles).
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 8;
__global__ __launch_bounds__(kScanThreads) void k_scan_totals(const int* __restrict__ plan, PlanLayout L, int nV, int* __restrict__ part) {
    __shared__ int wave_tot[kScanThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int* cnt = plan + L.cursor;
    const int i0 = (blockIdx.x * kScanThreads + tid) * kScanItems;
    int tot = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) tot += (i0 + j < nV) ? cnt[i0 + j] : 0;
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) wave_tot[wid] = tot;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += wave_tot[w];
        part[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(kScanThreads) void k_scan(int* __restrict__ plan, PlanLayout L, int nV, const int* __restrict__ part) {
    __shared__ int wave_tot[kScanThreads / 64];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int* cnt = plan + L.cursor;
    int* row_ptr = plan + L.row_ptr;
    {   // totals of the blocks before this one
        int c = 0;
        for (int b = tid; b < (int)blockIdx.x; b += kScanThreads) c += part[b];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0) wave_tot[wid] = c;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < kScanThreads / 64; ++w) t += wave_tot[w];
            carry_s = t;
        }
        __syncthreads();
    }
    const int carry = carry_s;
    int v[kScanItems];
    int tot = 0;
    const int i0 = (blockIdx.x * kScanThreads + tid) * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = (i0 + j < nV) ? cnt[i0 + j] : 0;
        tot += v[j];
    }
    int inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();  // (carry_s / wave_tot of the prefix pass have been read)
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wid; ++w) wave_base += wave_tot[w];
    int run = carry + wave_base + inc - tot;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (i0 + j < nV) {
            row_ptr[i0 + j] = run;
            cnt[i0 + j] = run;
        }
        run += v[j];
    }
    if (i0 <= nV - 1 && nV - 1 < i0 + kScanItems) row_ptr[nV] = run;  // (the thread that owns the last atom: run is the grand total)
    if (nV == 0 && blockIdx.x == 0 && tid == 0) row_ptr[0] = 0;
}

__global__ void k_fill(int* __restrict__ plan, PlanLayout L, int nE) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nE) {
        const int d = plan[L.dst + e];
        const int pos = atomicAdd(&plan[L.cursor + d], 1);
        plan[L.perm + pos] = e;
    }
}

// One thread per atom: put the (atomically filled, hence arbitrarily ordered) row into increasing
// edge id.  Rows are tiny (in-degree <= 4-6 for molecules), insertion sort in place.
__global__ void k_sort_rows(int* __restrict__ plan, PlanLayout L, int nV) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    int n = 0;
    if (v < nV) {
        const int b = plan[L.row_ptr + v];
        n = plan[L.row_ptr + v + 1] - b;
        int* row = plan + L.perm + b;
        for (int i = 1; i < n; ++i) {
            const int key = row[i];
            int j = i - 1;
            while (j >= 0 && row[j] > key) {
                row[j + 1] = row[j];
                --j;
            }
            row[j + 1] = key;
        }
    }
    int m = n;
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&plan[DMPNN_HDR_MAXDEG], m);
}

__global__ void k_inverse(int* __restrict__ plan, PlanLayout L, int nE) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nE) plan[L.inv + plan[L.perm + i]] = i;
}

// ---- row tiles of whole atoms ------------------------------------------------------------------
// Nominal stride B0 = BM - maxdeg + 1: atom v belongs to tile row_ptr[v] / B0, so a tile's rows start
// inside [t*B0, (t+1)*B0) and run at most maxdeg-1 rows past it: never more than BM rows, and no
// sequential packing pass is needed.  Trailing tiles (up to the launch bound) are empty.
struct TileGeom {
    int b0, n_tiles;  // n_tiles == 0: no fused tiling (no edges, or an in-degree the tiling cannot hold)
};
__device__ __forceinline__ TileGeom tile_geom(int maxdeg, int nE) {
    TileGeom g;
    const int md = maxdeg < 1 ? 1 : maxdeg;
    g.b0 = kFusedBM - md + 1;
    g.n_tiles = (md > kFusedMaxDeg || nE == 0) ? 0 : (nE + g.b0 - 1) / g.b0;
    return g;
}
__device__ __forceinline__ int tile_of(int row_start, const TileGeom& g) {
    const int t = row_start / g.b0;
    return t < g.n_tiles - 1 ? t : g.n_tiles - 1;
}
// thread `i` of `n_threads` cooperating threads writes its share of the two tables
__device__ __forceinline__ void write_tiles(int* __restrict__ plan, const PlanLayout& L, const int* __restrict__ row_ptr,
                                            int nV, int nE, const TileGeom& g, int i, int n_threads) {
    int* tile_row = plan + L.tile_row;
    int* tile_atom = plan + L.tile_atom;
    const int slots = (int)L.max_tiles + 2;
    if (g.n_tiles == 0) {
        for (int t = i; t < slots; t += n_threads) { tile_row[t] = nE; tile_atom[t] = nV; }
        return;
    }
    for (int v = i; v < nV; v += n_threads) {
        const int rs = row_ptr[v];
        const int tv = tile_of(rs, g);
        const int tp = v == 0 ? -1 : tile_of(row_ptr[v - 1], g);
        for (int t = tp + 1; t <= tv; ++t) { tile_row[t] = rs; tile_atom[t] = v; }
    }
    const int tl = nV > 0 ? tile_of(row_ptr[nV - 1], g) : -1;
    for (int t = tl + 1 + i; t < slots; t += n_threads) { tile_row[t] = nE; tile_atom[t] = nV; }
}

// CSR-row coordinates + tile tables + final header words (general path, after k_inverse).
// keep_mtiles: the molecule tiles came from the batch vector (dmpnn_tiles_large.hip), in the caller's edge order; the
// tile kernels read a FULL plan by rows, so a tile must also be the rows [row_ptr[first atom], row_ptr[end atom]) — which
// it is when the edges come in molecule order (collate.py:51-56): verified here for every table slot — and closed: every
// row's source atom inside the tile of its destination (with the reverse-edge invariants k_convert_count checks, the
// reverse row is then inside too).  A violation sets DMPNN_PLAN_NO_PIECE_TILES: the tile kernels return NaN, the
// per-step routes are untouched.
__global__ void k_rows_tiles(int* __restrict__ plan, PlanLayout L, int nV, int nE, int keep_mtiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_threads = gridDim.x * blockDim.x;
    const int maxdeg = plan[DMPNN_HDR_MAXDEG];
    const TileGeom g = tile_geom(maxdeg, nE);
    for (int r = i; r < nE; r += n_threads) {
        const int e = plan[L.perm + r];
        plan[L.srcp + r] = plan[L.src + e];
        plan[L.dstp + r] = plan[L.dst + e];
        plan[L.revp + r] = plan[L.inv + plan[L.rev + e]];
        plan[L.ident + r] = r;
    }
    write_tiles(plan, L, plan + L.row_ptr, nV, nE, g, i, n_threads);
    if (keep_mtiles) {
        const int* matom = plan + L.mtile_atom;
        const int* mrow = plan + L.mtile_row;
        const int* row_ptr = plan + L.row_ptr;
        const int nmt = plan[DMPNN_HDR_NMTILES];  // (not written by this kernel in this mode)
        int bad = 0;
        for (int t = i; t < (int)L.max_mtiles + 2; t += n_threads) {
            const int a = matom[t];
            if (a < 0 || a > nV || mrow[t] != row_ptr[a]) bad = 1;
        }
        if (nmt > 0) {
            for (int r = i; r < nE; r += n_threads) {
                const int e = plan[L.perm + r];
                const int d = plan[L.dst + e], sa = plan[L.src + e];
                int lo = 0, hi = nmt;  // last tile whose first atom is <= d (the table ends with the n_atoms sentinel)
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (matom[mid] <= d) lo = mid; else hi = mid;
                }
                if (sa < matom[lo] || sa >= matom[lo + 1]) bad = 1;
            }
        }
        if (bad) atomicOr(&plan[DMPNN_HDR_FLAGS], PLAN_NO_PIECE_TILES);
    } else {
        // no batch vector: piece tiles are built by the single-workgroup plan only — none here
        for (int t = i; t < (int)L.max_mtiles + 2; t += n_threads) { plan[L.mtile_row + t] = nE; plan[L.mtile_atom + t] = nV; }
    }
    if (i == 0) {
        plan[DMPNN_HDR_NTILES] = g.n_tiles;
        plan[DMPNN_HDR_TILE_STRIDE] = g.b0;
        if (!keep_mtiles) plan[DMPNN_HDR_NMTILES] = 0;
        atomicOr(&plan[DMPNN_HDR_FLAGS], (keep_mtiles ? 0 : PLAN_NO_PIECE_TILES) | (maxdeg > kFusedMaxDeg ? PLAN_HUGE_DEGREE : 0));
    }
}

// ---- single-workgroup plan for small batches (the launch-bound regime) -------------------------
constexpr int kSmallThreads = 1024;
constexpr int kSmallMaxAtoms = 6144;
constexpr int kSmallMaxEdges = 10240;
constexpr int kSmallItems = kSmallMaxAtoms / kSmallThreads;  // 6 counters per thread in the scan
constexpr int kSmallEPT = kSmallMaxEdges / kSmallThreads;    // 10 edges per thread at most
typedef unsigned short u16;


// In-place inclusive scan of a[0..n) (n <= kSmallItems * kSmallThreads) by the whole workgroup.
// MAX = false: sum, MAX = true: max.  Ends with a barrier.
template <bool MAX>
__device__ __forceinline__ void block_scan_inclusive(int* a, int n, int* wave_tot, int tid) {
    const int lane = tid & 63, wid = tid >> 6;
    int v[kSmallItems];
    const int i0 = tid * kSmallItems;
    int run = MAX ? INT_MIN : 0;
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
        const int x = (i0 + j < n) ? a[i0 + j] : (MAX ? INT_MIN : 0);
        run = MAX ? max(run, x) : run + x;
        v[j] = run;
    }
    int inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc = MAX ? max(inc, t) : inc + t;
    }
    __syncthreads();  // callers may still be reading wave_tot from a previous scan
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    int base = MAX ? INT_MIN : 0;
    for (int w = 0; w < wid; ++w) base = MAX ? max(base, wave_tot[w]) : base + wave_tot[w];
    const int prev = __shfl_up(inc, 1);
    const int excl = lane == 0 ? base : (MAX ? max(base, prev) : base + prev);
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j)
        if (i0 + j < n) a[i0 + j] = MAX ? max(excl, v[j]) : excl + v[j];
    __syncthreads();
}

// ---- piece tiles: row tiles made of WHOLE pieces (connected pieces, or molecules) ----------------
// Greedy packing of consecutive pieces into tiles of <= kMegaBM rows and <= kMegaBA atoms.  X[p] = first atom of
// piece p (X[np] = nV), row_of(p) = first row of piece p (row_of(np) = nE).  Y: int scratch of np + 1 entries.
// Returns the number of tiles, or -1 when a piece does not fit (the tables are then emptied).
template <class RowOf, class Other>
__device__ __forceinline__ int pack_pieces(int* __restrict__ plan, const PlanLayout& L, const int* X, int np, RowOf&& row_of,
                                           int* Y, int* bad_s, int* spill_s, int nV, int nE, int tid, Other&& other_work,
                                           long long* dbg, int n_st) {
    auto stamp = [&]() {
        if (dbg && tid == 0 && n_st < 12) dbg[n_st] = (long long)__builtin_readcyclecounter();
        ++n_st;
    };
    int* mrow = plan + L.mtile_row;
    int* matom = plan + L.mtile_atom;
    const int slots = (int)L.max_mtiles + 2;
    // next tile start (a piece index) for every piece; Y becomes the jump / mark word
    int nx[kSmallItems];
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
        const int p = tid + kSmallThreads * j;
        nx[j] = np;
        if (p < np) {
            const int v = X[p], r0 = row_of(p);
            int q = p + 1;  // pieces p .. q-1 fit
#if !defined(DMPNN_PACK_ROWS)
#define DMPNN_PACK_ROWS kMegaBM    /* (experiment builds: a smaller packing target — more, smaller tiles for the same batch) */
#endif
            if (X[q] - v > kMegaBA || row_of(q) - r0 > kMegaBM) {
                atomicAdd(spill_s, 1);  // one piece alone exceeds a tile: a tile of its own, for the generic in-kernel path
            } else {
                while (q < np && X[q + 1] - v <= kMegaBA && row_of(q + 1) - r0 <= DMPNN_PACK_ROWS) ++q;
            }
            nx[j] = q;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
        const int p = tid + kSmallThreads * j;
        if (p <= np) Y[p] = (p < np ? nx[j] : np);
    }
    __syncthreads();
    stamp();  // p4: next pointers
    // The chain of tile starts is walked per BLOCK of 64 consecutive pieces, one wave per block, all blocks
    // in parallel: a tile starts at every block start (at most one under-filled tile per block; the tile
    // before it simply ends there), lane l holds nx of piece base + l in a register, a hop is a v_readlane
    // (no LDS round trip), the chain nodes of the block are the set bits of a wave-uniform mask, their rank a
    // popcount on top of the tile count of the blocks before.
    constexpr int kMaxBlocks = kSmallMaxAtoms / 64 + 1;
    __shared__ int ntile_s;
    __shared__ int blk_cnt[kMaxBlocks];
    __shared__ unsigned long long blk_mask[kMaxBlocks];
    if (tid == 0) ntile_s = 0;
    __syncthreads();
    // (wave index through v_readfirstlane: hipcc takes tid >> 6 for divergent, and with it the block's base, the chain cursor and the mask —
    //  the hop below was 14 VALU / SALU instructions with hazard nops around a v_readfirstlane + v_readlane pair, ~65 cycles; with a uniform
    //  wave index the cursor and the mask live in scalar registers: profiles/r06_k0_scalar_walk.txt)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (np + 63) >> 6;
    const bool walk = *bad_s == 0;
    if (walk) {
        for (int blk = wave; blk < nblk; blk += kSmallThreads / 64) {
            const int base = blk * 64, p = base + lane;
            const int my_nx = p < np ? Y[p] : np;
            const int lim = (base + 64 < np ? base + 64 : np);
            unsigned long long mask = 0ull;
            int es = base;  // scalar chain cursor
            while (es < lim) {
                mask |= 1ull << (es - base);
                es = __builtin_amdgcn_readlane(my_nx, es - base);
            }
            if (lane == 0) { blk_mask[blk] = mask; blk_cnt[blk] = __popcll(mask); }
        }
    }
#if defined(DMPNN_K0_STAMPS2)
    stamp();  // a: this wave's chains walked
#endif
    __syncthreads();
#if defined(DMPNN_K0_STAMPS2)
    stamp();  // b: every block's chain walked
#endif
    if (walk) {
        for (int blk = wave; blk < nblk; blk += kSmallThreads / 64) {
            int rank = 0;
            for (int b2 = lane; b2 < blk; b2 += 64) rank += blk_cnt[b2];
            for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off);
            const unsigned long long mask = blk_mask[blk];
            const int p = blk * 64 + lane;
            const bool on = (mask >> lane) & 1ull;
            const int r = rank + __popcll(mask & ((1ull << lane) - 1ull));
            if (on && r < (int)L.max_mtiles) {
                mrow[r] = row_of(p);
                matom[r] = X[p];
            }
            if (blk == nblk - 1 && lane == 0) {
                const int total = rank + __popcll(mask);
                if (total > (int)L.max_mtiles) atomicOr(bad_s, 1);
                ntile_s = total;
            }
        }
    }
#if defined(DMPNN_K0_STAMPS2)
    stamp();  // c: ranks + the tiles' table words written
#endif
    other_work(tid, kSmallThreads);
#if defined(DMPNN_K0_STAMPS2)
    stamp();  // d: the row-tile tables emptied
#endif
    __syncthreads();
    stamp();  // p5: chain walk
    const int n_tiles = ntile_s;
    const bool bad = *bad_s != 0;
    if (bad) {
        for (int t = tid; t < slots; t += kSmallThreads) { mrow[t] = nE; matom[t] = nV; }
        return -1;
    }
    for (int t = n_tiles + tid; t < slots; t += kSmallThreads) { mrow[t] = nE; matom[t] = nV; }
    return n_tiles;
}

// Pieces from connectivity.  A cut after atom v is safe when no edge joins atoms <= v with atoms > v; with atoms of a molecule
// contiguous (data/collate.py:48-56) the cuts are the molecule boundaries (and fragment boundaries).
// Consecutive pieces are packed greedily into tiles of <= kMegaBM rows and <= kMegaBA atoms; the chain
// of tile starts is walked per block of 64 pieces, all blocks in parallel (a tile starts at every block
// start).  X, Y: int scratch of nV + 2 entries each; Y enters holding one byte per atom boundary
// (byte u != 0: an edge joins atoms < u with atoms >= u).
// Returns the number of tiles, or -1 when a piece does not fit (the tables are then emptied).
template <class Other>
__device__ __forceinline__ int build_piece_tiles(int* __restrict__ plan, const PlanLayout& L, const int* rowp, int* X,
                                                 int* Y, int* wave_tot, int* bad_s, int* spill_s, int nV, int nE, int tid,
                                                 Other&& other_work, long long* dbg = nullptr) {
    int n_st = 0;
    auto stamp = [&]() {
        if (dbg && tid == 0 && n_st < 12) dbg[n_st] = (long long)__builtin_readcyclecounter();
        ++n_st;
    };
    stamp();
    // piece starts = uncovered boundaries.  Y enters holding one covered-flag byte per atom boundary; a
    // ballot per 64 atoms turns them into start-bit words (behind the bytes), ranked by one wave scan over
    // the word popcounts.
    const int n_words = (nV + 31) >> 5;
    const unsigned char* covb = reinterpret_cast<const unsigned char*>(Y);
    unsigned* stw = reinterpret_cast<unsigned*>(Y + (nV + 3) / 4 + 1);  // [2 ceil(nV / 64)] start-bit words
    int* wrank = reinterpret_cast<int*>(stw) + 2 * ((nV + 63) >> 6);    // [n_words] start bits in the words before
    for (int u0 = (tid >> 6) * 64; u0 < nV; u0 += kSmallThreads) {
        const int u = u0 + (tid & 63);
        const unsigned long long m = __ballot(u < nV && covb[u] == 0);
        if ((tid & 63) == 0) { stw[u0 >> 5] = (unsigned)m; stw[(u0 >> 5) + 1] = (unsigned)(m >> 32); }
    }
    __syncthreads();
    const unsigned my_starts = tid < n_words ? stw[tid] : 0u;
    stamp();  // p1: start bits
    __shared__ int np_s;
    {   // kSmallMaxAtoms / 32 = 192 words: wave w scans words 64 w .. 64 w + 63, totals through LDS
        const int lane = tid & 63, w = tid >> 6;
        const int pc = __popc(my_starts);
        int inc = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[w] = inc;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w; ++k) base += wave_tot[k];
        if (tid < n_words) wrank[tid] = base + inc - pc;
        if (tid == n_words - 1) np_s = base + inc;
        if (n_words == 0 && tid == 0) np_s = 0;
    }
    __syncthreads();
    stamp();  // p2: start ranks
    const int np = np_s;
    if (tid < n_words) {                                       // X[p] = first atom of piece p (the cursors are dead)
        int p = wrank[tid];
        unsigned bits = my_starts;
        while (bits) {
            const int bpos = __ffs(bits) - 1;
            bits &= bits - 1u;
            X[p++] = tid * 32 + bpos;
        }
    }
    if (tid == 0) X[np] = nV;
    __syncthreads();
    stamp();  // p3: piece starts
    return pack_pieces(plan, L, X, np, [&](int p) { return rowp[X[p]]; }, Y, bad_s, spill_s, nV, nE, tid, other_work, dbg, n_st);
}

__global__ __launch_bounds__(kSmallThreads) void k_prepare_small(const int64_t* __restrict__ edge_index,
                                                                const int64_t* __restrict__ rev64,
                                                                int* __restrict__ plan, PlanLayout L, int nV, int nE,
                                                                int light, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    int n_stamp = 0;
    auto stamp = [&]() {
        if (dbg && threadIdx.x == 0 && n_stamp < 16) dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();
    // layout: cnt[nV + 2] int | row0[nV + 2] int | src16[nE] | dst16[nE] | rev16[nE] | perm16[nE] | inv16[nE] | Y[nV + 2] int
    // (the piece-tile phase reuses cnt as X)
    int* cnt = lds_i;
    int* rowp = lds_i + nV + 2;
    u16* src16 = reinterpret_cast<u16*>(rowp + nV + 2);
    u16* dst16 = src16 + nE;
    u16* rev16 = dst16 + nE;
    u16* perm16 = rev16 + nE;
    u16* inv16 = perm16 + nE;
    int* Ybuf = reinterpret_cast<int*>(lds_i) + 2 * (nV + 2) + (5 * nE + 1) / 2;  // [nV + 2] scratch of the piece tiles
    __shared__ int wave_tot[kSmallThreads / 64];
    __shared__ int flags_s, maxdeg_s, piece_bad_s, spill_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < nV; i += kSmallThreads) cnt[i] = 0;
    if (tid == 0) { flags_s = 0; maxdeg_s = 0; piece_bad_s = 0; spill_s = 0; }
    // light == 2, the TILE plan: only the piece-tile tables (+ header) — what an inference forward of the whole-forward
    // tile kernel reads; that kernel then takes src / dst / rev of a tile's edges straight from the caller's arrays
    // (its rows are the tile's edges in the caller's order: no sort, no CSR, no permutation) and checks on its own
    // that the tile is closed.  The boundary marks of the piece detection are set per bond while narrowing.
    const bool lean = light == 2;
    unsigned char* covb = reinterpret_cast<unsigned char*>(Ybuf);  // covb[u] != 0: an edge joins atoms < u with atoms >= u
    if (lean)
        for (int w = tid; w < (nV + 3) / 4 + 1; w += kSmallThreads) Ybuf[w] = 0;

    // phase 1: ONE batched read of the int64 arrays (all loads of a thread in flight together)
    int64_t s64[kSmallEPT], d64[kSmallEPT], r64[kSmallEPT];
#pragma unroll
    for (int j = 0; j < kSmallEPT; ++j) { s64[j] = 0; d64[j] = 0; r64[j] = 0; }
    if (nE > 0) {  // uniform: an edgeless batch has no (possibly null) index arrays to read
#pragma unroll
        for (int j = 0; j < kSmallEPT; ++j) {
            const int e = tid + kSmallThreads * j;
            const int ec = e < nE ? e : 0;
            s64[j] = edge_index[ec];
            d64[j] = edge_index[(int64_t)nE + ec];
            if (light != 2) r64[j] = rev64[ec];  // (a tile plan never looks at rev: the tile kernel checks it per tile)
        }
    }
    __syncthreads();  // cnt zeroed
    stamp();  // 1: int64 arrays loaded
    int bad = 0;
#pragma unroll
    for (int j = 0; j < kSmallEPT; ++j) {
        const int e = tid + kSmallThreads * j;
        int64_t s = s64[j], d = d64[j], r = r64[j];
        if (e < nE) {
            if (s < 0 || s >= nV || d < 0 || d >= nV || r < 0 || r >= nE) {
                bad |= PLAN_RANGE_ERROR | PLAN_ASYMMETRIC;
                s = s < 0 ? 0 : (s >= nV ? nV - 1 : s);
                d = d < 0 ? 0 : (d >= nV ? nV - 1 : d);
                r = r < 0 ? 0 : (r >= nE ? nE - 1 : r);
            }
            atomicAdd(&cnt[(int)d], 1);
            src16[e] = (u16)s;
            dst16[e] = (u16)d;
            rev16[e] = (u16)r;
        }
    }
    __syncthreads();
    if (lean) {  // boundary marks, once per bond (the direction with s < d), in one compact loop
        for (int e = tid; e < nE; e += kSmallThreads) {
            const int sa = src16[e], da = dst16[e];
            if (sa < da) {
                for (int u = sa + 1; u <= da; ++u) covb[u] = 1;  // (a bond that spans more atoms than a tile holds: its piece spills)
            }
        }
    }
    stamp();  // 2: narrowed + histogram
    // phase 1b: symmetric-graph invariants, from LDS
    if (!lean)
        for (int e = tid; e < nE; e += kSmallThreads) {
            const int r = rev16[e];
            if (rev16[r] != e || src16[r] != dst16[e] || dst16[r] != src16[e]) bad |= PLAN_ASYMMETRIC;
        }
    if (lean) bad &= ~PLAN_ASYMMETRIC;  // (not examined: the tile kernel's lean mode is exact for any rev map inside a tile)
    if (bad) atomicOr(&flags_s, bad);
    stamp();  // 3: validated
    // phase 2: exclusive scan of cnt[0..nV) (6 consecutive counters per thread)
    {
        int v[kSmallItems];
        int tot = 0;
        const int i0 = tid * kSmallItems;
#pragma unroll
        for (int j = 0; j < kSmallItems; ++j) {
            v[j] = (i0 + j < nV) ? cnt[i0 + j] : 0;
            tot += v[j];
        }
        int inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < wid; ++w) wave_base += wave_tot[w];
        int run = wave_base + inc - tot;
        int md = 0;
#pragma unroll
        for (int j = 0; j < kSmallItems; ++j) {
            if (i0 + j < nV) {
                rowp[i0 + j] = run;
                cnt[i0 + j] = run;
                if (!lean) plan[L.row_ptr + i0 + j] = run;
            }
            run += v[j];
            md = max(md, v[j]);
        }
        if (tid == kSmallThreads - 1) { rowp[nV] = run; if (!lean) plan[L.row_ptr + nV] = run; }
        for (int off = 32; off > 0; off >>= 1) md = max(md, __shfl_xor(md, off));
        if (lane == 0 && md > 0) atomicMax(&maxdeg_s, md);
    }
    __syncthreads();
    stamp();  // 4: scan
    // phase 3: fill rows (order inside a row is arbitrary here); the scratch of the piece tiles is zeroed for
    // the covered-boundary bytes of phase 4
    if (!lean) {
    for (int w = tid; w < (nV + 3) / 4 + 1; w += kSmallThreads) Ybuf[w] = 0;
    for (int e = tid; e < nE; e += kSmallThreads) {
        const int pos = atomicAdd(&cnt[dst16[e]], 1);
        perm16[pos] = (u16)e;
    }
    }
    __syncthreads();
    stamp();  // 5: fill
    // phase 4, one thread per atom: restore increasing edge id inside the row (the reference's summation
    // order), record the inverse permutation, and mark the atom boundaries the row's bonds reach across
    // (for the piece tiles).  Rows of <= 4 entries (molecules) are sorted in registers.
    if (!lean)
    for (int v = tid; v < nV; v += kSmallThreads) {
        const int b = rowp[v];
        const int n = rowp[v + 1] - b;
        u16* row = perm16 + b;
        int far = v;  // largest neighbour
        if (n <= 4) {
            const int last = n > 0 ? n - 1 : 0;
            unsigned e0 = row[0 <= last ? 0 : last], e1 = row[1 <= last ? 1 : last], e2 = row[2 <= last ? 2 : last], e3 = row[3 <= last ? 3 : last];
            e0 = n > 0 ? e0 : 0xffffu; e1 = n > 1 ? e1 : 0xffffu; e2 = n > 2 ? e2 : 0xffffu; e3 = n > 3 ? e3 : 0xffffu;
            unsigned t;
#define DMPNN_CSWAP(x, y) t = min(x, y); y = max(x, y); x = t
            DMPNN_CSWAP(e0, e1); DMPNN_CSWAP(e2, e3); DMPNN_CSWAP(e0, e2); DMPNN_CSWAP(e1, e3); DMPNN_CSWAP(e1, e2);
#undef DMPNN_CSWAP
            const unsigned ee[4] = {e0, e1, e2, e3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < n) {
                    row[k] = (u16)ee[k];
                    inv16[ee[k]] = (u16)(b + k);
                    far = max(far, (int)src16[ee[k]]);
                }
            }
        } else {
            for (int i = 1; i < n; ++i) {
                const u16 key = row[i];
                int j = i - 1;
                while (j >= 0 && row[j] > key) {
                    row[j + 1] = row[j];
                    --j;
                }
                row[j + 1] = key;
            }
            for (int k = 0; k < n; ++k) {
                inv16[row[k]] = (u16)(b + k);
                far = max(far, (int)src16[row[k]]);
            }
        }
        for (int u = v + 1; u <= far; ++u) covb[u] = 1;  // (a bond that spans more atoms than a tile holds: its piece spills)
    }
    __syncthreads();
    stamp();  // 6: sort
    stamp();  // 7: (inverse folded into phase 4)
    // phase 6 (runs on waves 1..15 while wave 0 walks the tile chain of phase 7): everything out.  A light
    // plan (forward of the fused routes only) skips the six arrays only the general route, the backward
    // pass and the tests read.
    TileGeom g = tile_geom(maxdeg_s, nE);
    if (lean) g.n_tiles = 0;  // no row tiles of the per-step fused route in a tile plan: its tables are emptied
    auto outputs = [&](int i0, int n_thr) {
        for (int i = i0; i < (lean ? 0 : nE); i += n_thr) {
            const int e = perm16[i];
            plan[L.perm + i] = e;
            plan[L.srcp + i] = src16[e];
            plan[L.revp + i] = inv16[rev16[e]];
            if (!light) {
                plan[L.src + i] = src16[i];
                plan[L.dst + i] = dst16[i];
                plan[L.rev + i] = rev16[i];
                plan[L.inv + i] = inv16[i];
                plan[L.dstp + i] = dst16[e];
                plan[L.ident + i] = i;
            }
        }
        write_tiles(plan, L, rowp, nV, nE, g, i0, n_thr);
    };
    stamp();  // 8: (outputs moved into phase 7)
    stamp();  // 9: (maxnbr folded into phase 5)
    // phase 7: piece tiles (scans by the whole workgroup, chain walk by wave 0 || outputs by waves 1..15)
    const int n_mtiles = build_piece_tiles(plan, L, rowp, cnt, Ybuf, wave_tot, &piece_bad_s, &spill_s, nV, nE, tid, outputs, dbg ? dbg + 16 : nullptr);
    stamp();  // 10: piece tiles
    if (tid < DMPNN_HDR_WORDS) {
        int v = 0;
        if (tid == DMPNN_HDR_FLAGS) v = flags_s | (maxdeg_s > kFusedMaxDeg ? PLAN_HUGE_DEGREE : 0) | (n_mtiles < 0 ? PLAN_NO_PIECE_TILES : 0) | (lean ? PLAN_TILES_ONLY : 0);
        if (tid == DMPNN_HDR_NMTILES) v = n_mtiles < 0 ? 0 : n_mtiles;
        if (tid == DMPNN_HDR_NSPILL) v = n_mtiles < 0 ? 0 : spill_s;
        if (tid == DMPNN_HDR_LIGHT) v = light;
        if (tid == DMPNN_HDR_MAXDEG) v = maxdeg_s;
        if (tid == DMPNN_HDR_NATOMS) v = nV;
        if (tid == DMPNN_HDR_NEDGES) v = nE;
        if (tid == DMPNN_HDR_NTILES) v = g.n_tiles;
        if (tid == DMPNN_HDR_TILE_STRIDE) v = g.b0;
        plan[tid] = v;
    }
}


// ---- tile plan from the batch vector (dmpnn_prepare_tiles with `batch`) -----------------------------
// BatchMolGraph.batch gives the molecule of every atom, non-decreasing, and collate concatenates the edges of the
// molecules in the same order (data/collate.py:48-62): molecule m owns the atoms lower_bound(batch, m) .. and the
// edges lower_bound(batch[dst[.]], m) ..  — two binary searches per molecule over arrays held in LDS, no histogram,
// no scan, no connectivity analysis.  Pieces = molecules.  What is NOT checked here (that every edge of a molecule's
// range has both atoms and its reverse inside) is checked by the tile kernel for every tile it runs.
__device__ __forceinline__ void tiles_batch_finish(int* fa, int* fe, int* Y, int nm, int* bad_sp, int* flags_sp, int* spill_sp, int* __restrict__ plan,
                                                   const PlanLayout& L, int nV, int nE, long long* dbg, int n_stamp, int* __restrict__ mol_bounds,
                                                   int n_mols_out);
__device__ __forceinline__ void prepare_tiles_batch_body(int* lds_i, const int64_t* __restrict__ edge_index,
                                                         const int64_t* __restrict__ batch,
                                                         int* __restrict__ plan, const PlanLayout& L, int nV, int nE,
                                                         long long* dbg, int* __restrict__ mol_bounds, int n_mols_out) {
    int n_stamp = 0;
    auto stamp = [&]() {
        if (dbg && threadIdx.x == 0 && n_stamp < 16) dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();
    int* fa = lds_i;                 // [nV + 2] first atom of molecule m
    int* fe = fa + nV + 2;           // [nV + 2] first edge of molecule m
    int* Y = fe + nV + 2;            // [nV + 2] next-tile pointers
    u16* bm = reinterpret_cast<u16*>(Y + nV + 2);  // [nV] molecule of an atom
    u16* mb = bm + nV + (nV & 1);                  // [nE] molecule of an edge (of its destination atom)
    __shared__ int bad_s, flags_s, nm_s, spill_s;
    const int tid = threadIdx.x;
    if (tid == 0) { bad_s = 0; flags_s = 0; nm_s = 0; spill_s = 0; }
    // phase 1: batch and dst in one batch of loads; batch -> LDS; then molecule of an edge = LDS lookup of its dst
    int64_t d64[kSmallEPT], b64[kSmallItems];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
        const int v = tid + kSmallThreads * j;
        b64[j] = batch[v < nV ? v : 0];
    }
#pragma unroll
    for (int j = 0; j < kSmallEPT; ++j) d64[j] = 0;
    if (nE > 0) {
#pragma unroll
        for (int j = 0; j < kSmallEPT; ++j) {
            const int e = tid + kSmallThreads * j;
            d64[j] = edge_index[(int64_t)nE + (e < nE ? e : 0)];
        }
    }
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
        const int v = tid + kSmallThreads * j;
        if (v < nV) {
            if (b64[j] < 0 || b64[j] >= nV) { bad |= PLAN_RANGE_ERROR; b64[j] = 0; }  // at most one molecule per atom
            bm[v] = (u16)b64[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSmallEPT; ++j) {
        const int e = tid + kSmallThreads * j;
        if (e < nE) {
            const bool ok = d64[j] >= 0 && d64[j] < nV;
            if (!ok) bad |= PLAN_RANGE_ERROR;
            mb[e] = bm[ok ? (int)d64[j] : 0];
        }
    }
    __syncthreads();
    stamp();  // 1: loaded, narrowed
    // phase 2: both must be non-decreasing (else the ranges below mean nothing: no tiles)
    for (int v = tid + 1; v < nV; v += kSmallThreads)
        if (bm[v] < bm[v - 1]) bad |= PLAN_NO_PIECE_TILES;
    for (int e = tid + 1; e < nE; e += kSmallThreads)
        if (mb[e] < mb[e - 1]) bad |= PLAN_NO_PIECE_TILES;
    if (bad) atomicOr(&flags_s, bad);
    if (tid == 0) nm_s = nV > 0 ? (int)bm[nV - 1] + 1 : 0;
    __syncthreads();
    stamp();  // 2: order checked
    const int nm = nm_s;
    // phase 3: first atom / first edge of every molecule (and the end markers at m = nm)
    for (int m = tid; m <= nm; m += kSmallThreads) {  // both searches in one loop: their LDS reads overlap
        int lo = 0, hi = nV, lo2 = 0, hi2 = nE;
        while (lo < hi || lo2 < hi2) {
            const int mid = (lo + hi) >> 1, mid2 = (lo2 + hi2) >> 1;
            const int x = lo < hi ? (int)bm[mid] : 0, y = lo2 < hi2 ? (int)mb[mid2] : 0;
            if (lo < hi) { if (x < m) lo = mid + 1; else hi = mid; }
            if (lo2 < hi2) { if (y < m) lo2 = mid2 + 1; else hi2 = mid2; }
        }
        fa[m] = lo;
        fe[m] = lo2;
    }
    if ((flags_s & (PLAN_NO_PIECE_TILES | PLAN_RANGE_ERROR)) && tid == 0) bad_s = 1;
    __syncthreads();
    stamp();  // 3: molecule ranges
    tiles_batch_finish(fa, fe, Y, nm, &bad_s, &flags_s, &spill_s, plan, L, nV, nE, dbg, n_stamp, mol_bounds, n_mols_out);
}

// ... the part behind the molecule ranges fa[0..nm] / fe[0..nm] (LDS): the optional bounds table, the greedy packing, the tables, the header
__device__ __forceinline__ void tiles_batch_finish(int* fa, int* fe, int* Y, int nm, int* bad_sp, int* flags_sp, int* spill_sp, int* __restrict__ plan,
                                                   const PlanLayout& L, int nV, int nE, long long* dbg, int n_stamp, int* __restrict__ mol_bounds,
                                                   int n_mols_out) {
    const int tid = threadIdx.x;
    auto stamp = [&]() {
        if (dbg && threadIdx.x == 0 && n_stamp < 16) dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    int& bad_s = *bad_sp; int& flags_s = *flags_sp; int& spill_s = *spill_sp;
    // (optional) the same ranges as the table the per-molecule aggregation reads (dmpnn_molagg.hip: first[n] | end[n] | flag) — a
    // training step that plans with this kernel does not launch k_mol_bounds for it
    if (mol_bounds) {
        for (int m = tid; m < n_mols_out; m += kSmallThreads) {
            mol_bounds[m] = m <= nm ? fa[m] : nV;
            mol_bounds[n_mols_out + m] = m + 1 <= nm ? fa[m + 1] : nV;
            mol_bounds[2 * n_mols_out + 4 + m] = 0;   // done[m]: set by the forward tile kernel where it writes the molecule's aggregate
        }
        if (tid == 0)   // 1: an id out of range (or beyond the caller's molecule count), 2: not sorted — any non-zero value poisons the aggregation
            mol_bounds[2 * n_mols_out] = ((flags_s & PLAN_RANGE_ERROR) || nm > n_mols_out ? 1 : 0) | ((flags_s & PLAN_NO_PIECE_TILES) ? 2 : 0);
    }
    // phase 4: greedy packing + the tables; the row tiles of the per-step fused route are emptied
    TileGeom g;
    g.b0 = kFusedBM; g.n_tiles = 0;
    auto rest = [&](int i0, int n_thr) { write_tiles(plan, L, fe, nV, nE, g, i0, n_thr); };
    const int n_mtiles = pack_pieces(plan, L, fa, nm, [&](int p) { return fe[p]; }, Y, &bad_s, &spill_s, nV, nE, tid, rest, dbg ? dbg + 16 : nullptr, 3);
    stamp();  // 4: packed
    if (tid < DMPNN_HDR_WORDS) {
        int v = 0;
        if (tid == DMPNN_HDR_FLAGS) v = (flags_s & PLAN_RANGE_ERROR) | (n_mtiles < 0 ? PLAN_NO_PIECE_TILES : 0) | PLAN_TILES_ONLY;
        if (tid == DMPNN_HDR_NMTILES) v = n_mtiles < 0 ? 0 : n_mtiles;
        if (tid == DMPNN_HDR_NSPILL) v = n_mtiles < 0 ? 0 : spill_s;
        if (tid == DMPNN_HDR_LIGHT) v = 2;
        if (tid == DMPNN_HDR_NATOMS) v = nV;
        if (tid == DMPNN_HDR_NEDGES) v = nE;
        if (tid == DMPNN_HDR_TILE_STRIDE) v = g.b0;
        plan[tid] = v;
    }
}
__global__ __launch_bounds__(kSmallThreads) void k_prepare_tiles_batch(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ batch,
                                                                      int* __restrict__ plan, PlanLayout L, int nV, int nE,
                                                                      long long* dbg, int* __restrict__ mol_bounds, int n_mols_out) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    prepare_tiles_batch_body(lds_i, edge_index, batch, plan, L, nV, nE, dbg, mol_bounds, n_mols_out);
}
// K0 and the pre-split of the forward's weights in ONE launch: workgroup 0 builds the tile table (11 us of one workgroup, 255 CUs
// idle), the other workgroups split the weight matrices (16 waves each, one wave per matrix row) — the split used to be a launch of
// its own (6 us) or, for frozen weights, cached between forwards on the strength of the tensors' autograd versions (a write through
// `param.data` went stale silently).  Now it costs nothing and is never stale: no cache.
__global__ __launch_bounds__(kSmallThreads) void k_prepare_tiles_batch_split(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ batch,
                                                                            int* __restrict__ plan, PlanLayout L, int nV, int nE,
                                                                            long long* dbg, int* __restrict__ mol_bounds, int n_mols_out,
                                                                            mega16::SplitArgs sp) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    if (blockIdx.x == 0) prepare_tiles_batch_body(lds_i, edge_index, batch, plan, L, nV, nE, dbg, mol_bounds, n_mols_out);
    else mega16::split_weights_wave(sp, ((int)blockIdx.x - 1) * (kSmallThreads / 64) + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}


// ---- K0 over SEVERAL workgroups in one launch (round 6) -------------------------------------------------------------------------
// The single-workgroup K0 above spends most of its 10 us pulling the batch vector and the destination row (110 KB of int64 at 512
// molecules) through ONE CU's fill path (~11 B/clk) and then searching them.  Here the launch is
//     blocks [0, n_bounds)            one thread per atom AND per edge: a molecule's first atom / first edge is where batch[.] /
//                                     batch[dst[.]] changes (dmpnn_tiles_large.hip's k_large_bounds) -> aoff[m], eoff[m] in the plan's
//                                     unused arrays, written THROUGH the L2 (agent-scope relaxed stores = `sc1`); every thread drains
//                                     its stores, the block's lane 0 then publishes an 8-byte {error bits, tag} word of the block;
//     blocks [n_bounds, last)         the forward's weight pre-split (as before: nothing depends on it inside the launch);
//     block  last                     waits for the n_bounds words (agent-scope relaxed loads, bounded spin), reads aoff / eoff with
//                                     agent-scope loads into LDS and goes on exactly as the single-workgroup kernel does from its
//                                     phase 3: packing, tables, header.  It then clears the words: the NEXT launch on this plan buffer
//                                     (also a replayed graph node with the same arguments) starts from "nothing published".
// The consumer is the LAST block: blocks are dispatched in index order, so every producer is resident or done when it starts to
// spin (the whole grid is ~100 workgroups on 256 CUs).  A word that never arrives (it cannot on a healthy launch) ends the spin
// after ~1 ms with PLAN_NO_PIECE_TILES: the tile kernel then writes NaN.  A fresh plan buffer holds garbage in the words: it passes
// for a published word only if its upper half equals the tag (2^-32 per block), and whatever table came of that is still checked by
// the tile kernel tile by tile (closure) — never silently wrong.
constexpr unsigned kBoundsTag = 0x6b30a11du;
constexpr int kBoundsMaxSpins = 4096;
// Round 6 (late): the offsets themselves carry the tag — aoff / eoff are 8-byte words {tag, offset}, written with ONE agent-scope store by the
// thread that finds the boundary, and the packing block polls THEM: a molecule's ranges are usable the moment they land, not a store drain,
// a barrier, a published block word and one more round trip later (packing block, cycles from its entry: ranges in LDS 7.9-8.5 k -> see
// profiles/r06_k0_tagged_offsets.txt).  The tag changes with every launch (a host counter), so a word left behind by an earlier launch on the
// same buffer never passes; the packing block clears what it read all the same.  The blocks' words {tag, error bits} are still published —
// behind the data, off the critical path — and the packing block looks at them LAST, before it ends: an error found there (an id out of range,
// ids not sorted) is patched into the header it has just written, so the tile kernel still poisons the batch.
struct MultiScratch { int64_t aoff, eoff, done, end; int n_bounds; unsigned tag; };
static MultiScratch multi_scratch(const PlanLayout& L, int64_t nV, int64_t nE) {
    MultiScratch S;
    S.n_bounds = (int)(((nV > nE ? nV : nE) + 1 + kSmallThreads - 1) / kSmallThreads);
    S.tag = 0u;
    int64_t o = (L.src + 1) & ~int64_t(1);   // (8-byte words)
    S.aoff = o; o += align4(2 * (nV + 2));
    S.eoff = o; o += align4(2 * (nV + 2));
    S.done = o; o += align4(2 * (int64_t)S.n_bounds);   // 8-byte words
    S.end = o;
    return S;
}
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_tagged(int* base, int m, unsigned tag, int v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base) + m, ((unsigned long long)tag << 32) | (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void tiles_bounds_block(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ batch, int* __restrict__ plan,
                                                   const MultiScratch& S, int nV, int nE, int b) {
    __shared__ int bad_blk;
    if (threadIdx.x == 0) bad_blk = 0;
    __syncthreads();
    const int i = b * kSmallThreads + (int)threadIdx.x;
    const int64_t last = batch[nV - 1];
    const int n_mols = last < 0 ? 0 : (last >= nV ? nV : (int)last + 1);
    int* aoff = plan + S.aoff;
    int* eoff = plan + S.eoff;
    const int64_t* dst = edge_index + nE;
    int bad = 0;
    auto clampm = [&](int64_t m) -> int { return m < 0 ? 0 : (m >= n_mols ? (n_mols > 0 ? n_mols - 1 : 0) : (int)m); };
    // all four reads of a thread in flight together; the molecule of an edge is a dependent read of the (L2-resident) batch vector
    const int64_t bm = i < nV ? batch[i] : 0, bp = (i > 0 && i < nV) ? batch[i - 1] : -1;
    int64_t d = (i < nE) ? dst[i] : 0, dp = (i > 0 && i <= nE && nE > 0) ? dst[i - 1] : 0;
    if (i < nE && (d < 0 || d >= nV)) { bad |= PLAN_RANGE_ERROR; d = 0; }
    if (dp < 0 || dp >= nV) dp = 0;   // (flagged by the thread that owns that edge)
    const int64_t em64 = i < nE ? batch[d] : 0, ep64 = (i > 0 && i <= nE && nE > 0) ? batch[dp] : -1;
    if (i < nV) {
        if (bm < 0 || bm >= nV) bad |= PLAN_RANGE_ERROR;   // at most one molecule per atom
        const int m = clampm(bm), pm = i > 0 ? clampm(bp) : -1;
        if (m < pm) bad |= PLAN_NO_PIECE_TILES;            // not non-decreasing: the ranges mean nothing
        for (int mm = pm + 1; mm <= m; ++mm) st_tagged(aoff, mm, S.tag, i);   // (empty unless i is a boundary; the boundary thread fills a gap of empty molecules)
    } else if (i == nV) {
        st_tagged(aoff, n_mols, S.tag, nV);
    }
    if (i < nE) {
        const int m = clampm(em64), pm = i > 0 ? clampm(ep64) : -1;
        if (m < pm) bad |= PLAN_NO_PIECE_TILES;
        for (int mm = pm + 1; mm <= m; ++mm) st_tagged(eoff, mm, S.tag, i);
    } else if (i == nE) {
        const int pm = nE > 0 ? clampm(ep64) : -1;
        for (int mm = pm + 1; mm <= n_mols; ++mm) st_tagged(eoff, mm, S.tag, nE);   // molecules behind the last edge, and the end marker
    }
    if (bad) atomicOr(&bad_blk, bad);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores have left
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(plan + S.done) + b, ((unsigned long long)S.tag << 32) | (unsigned)bad_blk,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void tiles_pack_block(int* lds_i, const int64_t* __restrict__ batch, int* __restrict__ plan, const PlanLayout& L,
                                                 const MultiScratch& S, int nV, int nE, long long* dbg, int* __restrict__ mol_bounds, int n_mols_out) {
    int n_stamp = 0;
    auto stamp = [&]() {
        if (dbg && threadIdx.x == 0 && n_stamp < 16) dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();
    int* fa = lds_i;                 // [nV + 2] first atom of molecule m
    int* fe = fa + nV + 2;           // [nV + 2] first edge of molecule m
    int* Y = fe + nV + 2;            // [nV + 2] next-tile pointers
    __shared__ int bad_s, flags_s, spill_s;
    const int tid = threadIdx.x;
    if (tid == 0) { bad_s = 0; flags_s = 0; spill_s = 0; }
    const int64_t last = batch[nV - 1];
    const int nm = last < 0 ? 0 : (last >= nV ? nV : (int)last + 1);
    __syncthreads();
    unsigned long long* done = reinterpret_cast<unsigned long long*>(plan + S.done);
    {   // every molecule's first atom / first edge as soon as its word lands (bounded spin: a word that never comes — ids not sorted — is an error)
        unsigned long long* a64 = reinterpret_cast<unsigned long long*>(plan + S.aoff);
        unsigned long long* e64 = reinterpret_cast<unsigned long long*>(plan + S.eoff);
        for (int m = tid; m <= nm; m += kSmallThreads) {
            unsigned long long wa = 0ull, we = 0ull;
            int spins = 0;
            for (;;) {
                wa = __hip_atomic_load(a64 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                we = __hip_atomic_load(e64 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((unsigned)(wa >> 32) == S.tag && (unsigned)(we >> 32) == S.tag) || ++spins >= kBoundsMaxSpins) break;
                __builtin_amdgcn_s_sleep(2);
            }
            const bool ok = (unsigned)(wa >> 32) == S.tag && (unsigned)(we >> 32) == S.tag;
            if (!ok) atomicOr(&flags_s, PLAN_NO_PIECE_TILES);
            fa[m] = ok ? (int)(unsigned)wa : 0;
            fe[m] = ok ? (int)(unsigned)we : 0;
            __hip_atomic_store(a64 + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nothing published: the next launch's start state)
            __hip_atomic_store(e64 + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    stamp();  // 1: every range has landed
    __syncthreads();
    stamp();  // 2: ranges in LDS
    {   // whatever was read must at least be a pair of non-decreasing offset lists inside the arrays
        int bad = 0;
        for (int m = tid; m <= nm; m += kSmallThreads) {
            const int a = fa[m], e = fe[m];
            if (a < 0 || a > nV || e < 0 || e > nE) bad = 1;
            if (m > 0 && (fa[m - 1] > a || fe[m - 1] > e)) bad = 1;
        }
        if (bad) atomicOr(&flags_s, PLAN_NO_PIECE_TILES);
    }
    __syncthreads();
    if ((flags_s & (PLAN_NO_PIECE_TILES | PLAN_RANGE_ERROR)) && tid == 0) bad_s = 1;
    __syncthreads();
    stamp();  // 3: molecule ranges
    // (a first look at the blocks' words NOW: the load is in flight under the packing, and a word that is already there — the blocks publish
    //  ~1 k cycles behind their data — costs the end of the kernel no round trip)
    unsigned long long w_early = 0ull;
    if (tid < S.n_bounds) w_early = __hip_atomic_load(done + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tiles_batch_finish(fa, fe, Y, nm, &bad_s, &flags_s, &spill_s, plan, L, nV, nE, dbg, n_stamp, mol_bounds, n_mols_out);
    // ... and what the bounds blocks found wrong, LAST: their words are behind their data (normally here long ago)
    __shared__ int late_s;
    if (tid == 0) late_s = 0;
    __syncthreads();
    if (tid < S.n_bounds) {
        unsigned long long w = w_early;
        int spins = 0;
        while ((unsigned)(w >> 32) != S.tag && ++spins < kBoundsMaxSpins) {
            __builtin_amdgcn_s_sleep(8);
            w = __hip_atomic_load(done + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((unsigned)(w >> 32) != S.tag) atomicOr(&late_s, PLAN_NO_PIECE_TILES);
        else if ((unsigned)w) atomicOr(&late_s, (int)((unsigned)w & (PLAN_NO_PIECE_TILES | PLAN_RANGE_ERROR)));
        __hip_atomic_store(done + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0 && (late_s & ~flags_s)) {   // (an error the header does not carry yet: this block wrote the header, program order)
        plan[DMPNN_HDR_FLAGS] |= late_s & (PLAN_NO_PIECE_TILES | PLAN_RANGE_ERROR);
        if (late_s & PLAN_NO_PIECE_TILES) { plan[DMPNN_HDR_NMTILES] = 0; plan[DMPNN_HDR_NSPILL] = 0; }
        if (mol_bounds) mol_bounds[2 * n_mols_out] |= ((late_s & PLAN_RANGE_ERROR) ? 1 : 0) | ((late_s & PLAN_NO_PIECE_TILES) ? 2 : 0);
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_prepare_tiles_batch_multi(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ batch,
                                                                            int* __restrict__ plan, PlanLayout L, MultiScratch S, int nV, int nE,
                                                                            long long* dbg, int* __restrict__ mol_bounds, int n_mols_out,
                                                                            mega16::SplitArgs sp) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    const int b = (int)blockIdx.x;
    if (b < S.n_bounds) tiles_bounds_block(edge_index, batch, plan, S, nV, nE, b);
    else if (b + 1 < (int)gridDim.x) mega16::split_weights_wave(sp, (b - S.n_bounds) * (kSmallThreads / 64) + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
    else tiles_pack_block(lds_i, batch, plan, L, S, nV, nE, dbg, mol_bounds, n_mols_out);
}

// The loader's tile table (dmpnn_prepare_tiles_from_table) with the weight pre-split in the same launch: workgroup 0 copies / checks the
// table, the others split — the "K0 for free" path used to pay a k_split_weights launch per forward once the weight cache was gone
// (round-4 VERDICT weak #6: 40.4 -> 46.7 us at 512 molecules).
__global__ __launch_bounds__(256) void k_tiles_from_table_split(const int* __restrict__ tile_row, const int* __restrict__ tile_atom, int n_tiles,
                                                                int nV, int nE, int* __restrict__ plan, PlanLayout L, mega16::SplitArgs sp) {
    if (blockIdx.x == 0) tiles_from_table_body(tile_row, tile_atom, n_tiles, nV, nE, plan, L);
    else mega16::split_weights_wave(sp, ((int)blockIdx.x - 1) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

}  // namespace

int launch_tiles_from_table_split(const int* tile_row, const int* tile_atom, int64_t n_tiles, int64_t nV, int64_t nE, int* plan, hipStream_t s,
                                  const dmpnn_fwd_args* split_for, bool* did_split) {
    *did_split = false;
    mega16::SplitArgs sp;
    if (!split_for || !(split_for->flags & DMPNN_F_MEGA) || !(split_for->flags & DMPNN_F_SPLIT16) || (split_for->flags & DMPNN_F_WSPLIT_READY) ||
        !mega16_split_args(*split_for, &sp))
        return DMPNN_OK;   // (nothing to split here: the caller launches the plain table kernel)
    const PlanLayout L = plan_layout(nV, nE);
    const int waves = ((sp.N + 15) & ~15) * sp.n_jobs;
    hipLaunchKernelGGL(k_tiles_from_table_split, dim3((unsigned)(1 + (waves + 3) / 4)), dim3(256), 0, s, tile_row, tile_atom, (int)n_tiles, (int)nV,
                       (int)nE, plan, L, sp);
    DMPNN_CHECK_LAUNCH("k_tiles_from_table_split");
    *did_split = true;
    return DMPNN_OK;
}

// bytes of LDS of k_prepare_tiles_batch
static size_t tiles_batch_lds_bytes(int64_t nV, int64_t nE) {
    return ((size_t)(3 * (nV + 2)) * 4 + (size_t)(nV + 1 + nE + 2) * 2 + 31) & ~size_t(15);
}

int launch_prepare_tiles_batch(const int64_t* edge_index, const int64_t* batch, int64_t nV64, int64_t nE64, int* plan, hipStream_t s,
                               int* mol_bounds, int64_t n_mols, const dmpnn_fwd_args* split_for, bool* did_split) {
    const int nV = (int)nV64, nE = (int)nE64;
    const PlanLayout L = plan_layout(nV, nE);
    if (did_split) *did_split = false;
    mega16::SplitArgs sp;
    {   // K0 over several workgroups (k_prepare_tiles_batch_multi), with or without the weight pre-split riding in the launch
        static const bool multi_off = [] { const char* e = getenv("DMPNN_K0_SINGLE"); return e && atoi(e) != 0; }();
        MultiScratch S = multi_scratch(L, nV, nE);
        static std::atomic<unsigned> launch_no{0u};
        S.tag = kBoundsTag ^ (launch_no.fetch_add(1u, std::memory_order_relaxed) * 0x9E3779B1u);   // (never 0 twice in a row; a replayed graph node keeps its own)
        if (S.tag == 0u) S.tag = kBoundsTag;
        if (!multi_off && nV > 0 && S.end <= L.tile_row) {
            const bool split = split_for && did_split && (split_for->flags & DMPNN_F_MEGA) && (split_for->flags & DMPNN_F_SPLIT16) &&
                               !(split_for->flags & DMPNN_F_WSPLIT_READY) && mega16_split_args(*split_for, &sp);
            if (!split) memset(&sp, 0, sizeof(sp));
            static bool attr3_set = false;
            if (!attr3_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare_tiles_batch_multi),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
                if (e != hipSuccess) {
                    set_error("hipFuncSetAttribute(k_prepare_tiles_batch_multi): %s", hipGetErrorString(e));
                    return DMPNN_EHIP;
                }
                attr3_set = true;
            }
            const int waves = split ? ((sp.N + 15) & ~15) * sp.n_jobs : 0, wpb = kSmallThreads / 64;
            const size_t lds = ((size_t)(3 * (nV + 2)) * 4 + 31) & ~size_t(15);
            hipLaunchKernelGGL(k_prepare_tiles_batch_multi, dim3((unsigned)(S.n_bounds + (waves + wpb - 1) / wpb + 1)), dim3(kSmallThreads), lds, s,
                               edge_index, batch, plan, L, S, nV, nE, g_debug_stamps ? g_debug_stamps + 32 : nullptr, mol_bounds, (int)n_mols, sp);
            DMPNN_CHECK_LAUNCH("k_prepare_tiles_batch_multi");
            if (split) *did_split = true;
            return DMPNN_OK;
        }
    }
    if (split_for && did_split && (split_for->flags & DMPNN_F_MEGA) && (split_for->flags & DMPNN_F_SPLIT16) && !(split_for->flags & DMPNN_F_WSPLIT_READY) &&
        mega16_split_args(*split_for, &sp)) {
        static bool attr2_set = false;
        if (!attr2_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare_tiles_batch_split),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(k_prepare_tiles_batch_split): %s", hipGetErrorString(e));
                return DMPNN_EHIP;
            }
            attr2_set = true;
        }
        const int waves = ((sp.N + 15) & ~15) * sp.n_jobs, wpb = kSmallThreads / 64;
        hipLaunchKernelGGL(k_prepare_tiles_batch_split, dim3((unsigned)(1 + (waves + wpb - 1) / wpb)), dim3(kSmallThreads), tiles_batch_lds_bytes(nV, nE), s,
                           edge_index, batch, plan, L, nV, nE, g_debug_stamps ? g_debug_stamps + 32 : nullptr, mol_bounds, (int)n_mols, sp);
        DMPNN_CHECK_LAUNCH("k_prepare_tiles_batch_split");
        *did_split = true;
        return DMPNN_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare_tiles_batch),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(k_prepare_tiles_batch): %s", hipGetErrorString(e));
            return DMPNN_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k_prepare_tiles_batch, dim3(1), dim3(kSmallThreads), tiles_batch_lds_bytes(nV, nE), s, edge_index, batch, plan,
                       L, nV, nE, g_debug_stamps ? g_debug_stamps + 32 : nullptr, mol_bounds, (int)n_mols);
    DMPNN_CHECK_LAUNCH("k_prepare_tiles_batch");
    return DMPNN_OK;
}

// (the scan's per-block scratch moves to the row-tile table in that mode: it must hold one int per scan block)
bool prepare_can_keep_mtiles(int64_t nV, int64_t nE) {
    const PlanLayout L = plan_layout(nV, nE);
    return (nV + kScanThreads * kScanItems - 1) / (kScanThreads * kScanItems) <= L.max_tiles + 2;
}

int launch_prepare(const int64_t* edge_index, const int64_t* rev, int64_t nV64, int64_t nE64,
                   int* plan, int light, hipStream_t s, bool keep_mtiles) {
    const int nV = (int)nV64, nE = (int)nE64;
    const PlanLayout L = plan_layout(nV, nE);
    if (small_plan_fits(nV, nE)) {
        const size_t lds = small_plan_lds_bytes(nV, nE);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare_small),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(k_prepare_small): %s", hipGetErrorString(e));
                return DMPNN_EHIP;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL(k_prepare_small, dim3(1), dim3(kSmallThreads), lds, s, edge_index, rev, plan, L, nV, nE, light,
                           g_debug_stamps ? g_debug_stamps + 32 : nullptr);
        DMPNN_CHECK_LAUNCH("k_prepare_small");
        return DMPNN_OK;
    }
    // (beyond the single-workgroup plan: always the full plan, whatever `light` asked for)
    {
        const int64_t n = nV > DMPNN_HDR_WORDS ? nV : DMPNN_HDR_WORDS;
        int grid = (int)((n + kBlock - 1) / kBlock);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(k_plan_init, dim3(grid), dim3(kBlock), 0, s, plan, L.cursor, nV, nE, keep_mtiles ? 1 : 0);
        DMPNN_CHECK_LAUNCH("k_plan_init");
    }
    if (nE > 0) {
        const int grid = (nE + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_convert_count, dim3(grid), dim3(kBlock), 0, s, edge_index, rev, plan, L, nV, nE);
        DMPNN_CHECK_LAUNCH("k_convert_count");
    }
    {
        const int nblk = nV > 0 ? (nV + kScanThreads * kScanItems - 1) / (kScanThreads * kScanItems) : 1;
        // scratch: one int per block, in a table k_rows_tiles (re)writes at the end
        int* part = plan + (keep_mtiles ? L.tile_row : L.mtile_row);
        if (nblk > 1) {
            hipLaunchKernelGGL(k_scan_totals, dim3(nblk), dim3(kScanThreads), 0, s, plan, L, nV, part);
            DMPNN_CHECK_LAUNCH("k_scan_totals");
        }
        hipLaunchKernelGGL(k_scan, dim3(nblk), dim3(kScanThreads), 0, s, plan, L, nV, part);
        DMPNN_CHECK_LAUNCH("k_scan");
    }
    if (nE > 0) {
        const int grid = (nE + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_fill, dim3(grid), dim3(kBlock), 0, s, plan, L, nE);
        DMPNN_CHECK_LAUNCH("k_fill");
        const int gridv = (nV + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_sort_rows, dim3(gridv), dim3(kBlock), 0, s, plan, L, nV);
        DMPNN_CHECK_LAUNCH("k_sort_rows");
        hipLaunchKernelGGL(k_inverse, dim3(grid), dim3(kBlock), 0, s, plan, L, nE);
        DMPNN_CHECK_LAUNCH("k_inverse");
    }
    {
        const int64_t n = nV > nE ? nV : nE;
        int grid = (int)((n + kBlock - 1) / kBlock);
        if (grid < 1) grid = 1;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_rows_tiles, dim3(grid), dim3(kBlock), 0, s, plan, L, nV, nE, keep_mtiles ? 1 : 0);
        DMPNN_CHECK_LAUNCH("k_rows_tiles");
    }
    return DMPNN_OK;
}

}  // namespace dmpnn
