// Shared declarations for the gfx950 D-MPNN engine (internal; the public boundary is include/dmpnn.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dmpnn.h"

namespace dmpnn {

namespace mega16 { struct SplitArgs; }   // (argument block of the weight pre-split: dmpnn_mega16_impl.hpp)

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void count_launch(const char* name);

#define DMPNN_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::dmpnn::set_error(__VA_ARGS__);       \
            return DMPNN_EINVAL;                   \
        }                                          \
    } while (0)

#define DMPNN_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::dmpnn::set_error("launch of %s failed: %s", name, hipGetErrorString(e__)); \
            return DMPNN_EHIP;                                                           \
        }                                                                                \
        ::dmpnn::count_launch(name);                                                     \
    } while (0)

#define DMPNN_TRY(expr)               \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != DMPNN_OK) return rc__; \
    } while (0)

// ---- plan layout ----------------------------------------------------------------------------
// One int32 blob, every array on a 16-byte boundary:
//   hdr[16] | src[E] dst[E] rev[E] row_ptr[V+1] perm[E] cursor[V]          original edge ids
//           | inv[E] srcp[E] dstp[E] revp[E] ident[E]                      CSR-row coordinates
//           | tile_row[T+2] tile_atom[T+2]                                 row tiles of whole atoms
//           | mtile_row[MT+2] mtile_atom[MT+2]                             row tiles of whole connected pieces
// CSR-row coordinates: row i is the edge perm[i]; rows of one destination atom are contiguous and in
// increasing edge id.  srcp/dstp/revp are src/dst/rev expressed in rows (revp[i] = inv[rev[perm[i]]]),
// ident[i] = i.  The fused forward keeps every edge tensor in row order, so a row tile of the
// contraction holds WHOLE destination atoms and the segment sums are formed in its epilogue.
constexpr int kFusedBM = 48;        // rows of a fused tile (RT = 3)
constexpr int kFusedMaxDeg = 24;    // largest in-degree the tiling supports ((BM + 1) / 2)
constexpr int kFusedMinB0 = kFusedBM - kFusedMaxDeg + 1;  // 25: smallest nominal tile stride
constexpr int kMegaBM = 48;         // rows of a piece tile (whole connected pieces = molecules)
constexpr int kMegaBA = 32;         // atoms of a piece tile (rows of the finalize contraction, RT = 2)
struct PlanLayout {
    int64_t src, dst, rev, row_ptr, perm, cursor, inv, srcp, dstp, revp, ident, tile_row, tile_atom, mtile_row, mtile_atom, words;
    int64_t max_tiles, max_mtiles;
};
inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }
inline int64_t fused_max_tiles(int64_t nE) { return (nE + kFusedMinB0 - 1) / kFusedMinB0 + 1; }
// greedy packing: two consecutive piece tiles cannot be merged, so together they exceed a limit; plus one
// under-filled tile per block of 64 pieces (build_piece_tiles)
inline int64_t mega_max_tiles(int64_t nV, int64_t nE) { return 2 * (nE / kMegaBM + nV / kMegaBA) + 4 + nV / 64 + 1; }
inline PlanLayout plan_layout(int64_t nV, int64_t nE) {
    PlanLayout L;
    int64_t o = DMPNN_HDR_WORDS;
    L.src = o; o += align4(nE);
    L.dst = o; o += align4(nE);
    L.rev = o; o += align4(nE);
    L.row_ptr = o; o += align4(nV + 1);
    L.perm = o; o += align4(nE);
    L.cursor = o; o += align4(nV);
    L.inv = o; o += align4(nE);
    L.srcp = o; o += align4(nE);
    L.dstp = o; o += align4(nE);
    L.revp = o; o += align4(nE);
    L.ident = o; o += align4(nE);
    L.max_tiles = fused_max_tiles(nE);
    L.tile_row = o; o += align4(L.max_tiles + 2);
    L.tile_atom = o; o += align4(L.max_tiles + 2);
    L.max_mtiles = mega_max_tiles(nV, nE);
    L.mtile_row = o; o += align4(L.max_mtiles + 2);
    L.mtile_atom = o; o += align4(L.max_mtiles + 2);
    L.words = o;
    return L;
}

enum : int { PLAN_ASYMMETRIC = 1, PLAN_RANGE_ERROR = 2, PLAN_HUGE_DEGREE = 4, PLAN_NO_PIECE_TILES = 8,
             PLAN_TILES_ONLY = 16 };  // tile plan (dmpnn_prepare_tiles): no CSR arrays, only the piece-tile tables
// graphs the fused (row-tiled) forward cannot represent: its kernels poison their output with NaN
constexpr int kPlanNoFuse = PLAN_ASYMMETRIC | PLAN_RANGE_ERROR | PLAN_HUGE_DEGREE | PLAN_TILES_ONLY;
// what the whole-forward tile kernel cannot take when it reads a tile plan (its own per-tile checks do the rest)
constexpr int kPlanNoMegaLean = PLAN_RANGE_ERROR | PLAN_NO_PIECE_TILES;
// graphs the whole-forward tile kernel cannot take (a connected piece larger than a tile, or no piece tiles built)
constexpr int kPlanNoMega = kPlanNoFuse | PLAN_NO_PIECE_TILES;

// Device view of a plan (pointers into the blob).
struct PlanView {
    const int* hdr;
    const int* src;
    const int* dst;
    const int* rev;
    const int* row_ptr;
    const int* perm;
};
inline PlanView plan_view(const void* plan, int64_t nV, int64_t nE) {
    const int* p = static_cast<const int*>(plan);
    PlanLayout L = plan_layout(nV, nE);
    return PlanView{p, p + L.src, p + L.dst, p + L.rev, p + L.row_ptr, p + L.perm};
}
// The same graph in CSR-row coordinates (perm = identity): what the fused path's kept tensors use.
inline PlanView plan_view_rows(const void* plan, int64_t nV, int64_t nE) {
    const int* p = static_cast<const int*>(plan);
    PlanLayout L = plan_layout(nV, nE);
    return PlanView{p, p + L.srcp, p + L.dstp, p + L.revp, p + L.row_ptr, p + L.ident};
}

// ---- dropout mask: a counter-based hash (restated in oracle/dropout_hash.py) -----------------------------------------
// keep(seed, site, row, col) = mix32(...) >= thr,  thr = floor(p 2^32).  Two rounds of a multiply-xorshift mixer over a key
// that is unique per element (row * 1024 + col: d_h <= 1024), the 64-bit seed folded in before and between the rounds.
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned seed_lo, unsigned seed_hi, unsigned site, unsigned row, unsigned col) {
    unsigned h = (row * 1024u + col) ^ seed_lo ^ (site * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h += seed_hi; h *= 0x846CA68Bu; h ^= h >> 16;
    h *= 0x9E3779B1u; h ^= h >> 15;
    return h;
}
inline unsigned drop_threshold(float p) {  // floor(p 2^32), p in (0, 1)
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}

// ---- kernel arguments: one round trip -------------------------------------------------------
// A kernel whose argument block spans several 64-byte lines reads them lazily — one scalar load next to each first use, each a
// cold miss behind the branch before it (the block was written by the host a moment ago: the lines are in HBM, not in any cache).
// Touching every line at entry turns that chain into ONE round trip; the later loads hit the scalar cache.  (The tile kernels
// run one workgroup per CU at the headline size: nothing else hides that chain — profiles/r03_kernarg_warm.txt.)
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int L = (BYTES + 63) / 64;
    static_assert(L <= 12, "argument block larger than the warm-up covers");
    int d[12];
#define DMPNN_WARM(i) if constexpr (L > i) asm volatile("s_load_dword %0, %1, %2" : "=s"(d[i]) : "s"(kp), "n"(i * 64));
    DMPNN_WARM(0) DMPNN_WARM(1) DMPNN_WARM(2) DMPNN_WARM(3) DMPNN_WARM(4) DMPNN_WARM(5)
    DMPNN_WARM(6) DMPNN_WARM(7) DMPNN_WARM(8) DMPNN_WARM(9) DMPNN_WARM(10) DMPNN_WARM(11)
#undef DMPNN_WARM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- activations ----------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float z, int act, float slope) {
    switch (act) {
        case DMPNN_ACT_RELU: return z < 0.f ? 0.f : z;
        case DMPNN_ACT_LEAKYRELU:
        case DMPNN_ACT_PRELU: return z > 0.f ? z : slope * z;
        case DMPNN_ACT_TANH: return tanhf(z);
        case DMPNN_ACT_ELU: return z > 0.f ? z : expm1f(z);
        default: return z;
    }
}
// The transcendental activations out of line: a fully unrolled epilogue that inlines tanhf / expm1f per element grows by
// tens of KB, and a workgroup that runs its code once pays for code size in instruction fetches (the tile kernels:
// 27 % of their cycles).  ReLU-class activations are a compare + select in line.
__device__ __noinline__ float apply_act_slow(float z, int act) {
    return act == DMPNN_ACT_TANH ? tanhf(z) : (z > 0.f ? z : expm1f(z));
}
__device__ __forceinline__ float apply_act_small(float z, int act, float slope) {
    if (act == DMPNN_ACT_TANH || act == DMPNN_ACT_ELU) return apply_act_slow(z, act);
    const float neg = act == DMPNN_ACT_NONE ? 1.f : (act == DMPNN_ACT_RELU ? 0.f : slope);
    return (z > 0.f ? z : neg * z) + 0.f;
}
__device__ __forceinline__ float4 apply_act4(float4 z, int act, float slope) {
    if (act == DMPNN_ACT_NONE) return z;
    return make_float4(apply_act(z.x, act, slope), apply_act(z.y, act, slope),
                       apply_act(z.z, act, slope), apply_act(z.w, act, slope));
}
// d tau / dz given pre-activation z (when available) or output y.
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case DMPNN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DMPNN_ACT_LEAKYRELU: return y > 0.f ? 1.f : slope;  // slope > 0: sign(y) == sign(z)
        case DMPNN_ACT_TANH: return 1.f - y * y;
        case DMPNN_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        default: return 1.f;
    }
}

// ---- host launchers implemented in the .hip files -------------------------------------------
// keep_mtiles (batches beyond the single-workgroup plan only): launch_prepare_tiles_large wrote the molecule tiles of
// this plan just before on the same stream — the full plan keeps and verifies them instead of writing "no piece tiles"
int launch_prepare(const int64_t* edge_index, const int64_t* rev, int64_t nV, int64_t nE, int* plan,
                   int light, hipStream_t s, bool keep_mtiles = false);
bool prepare_can_keep_mtiles(int64_t nV, int64_t nE);
// The piece-tile tables of a tile plan from a table the LOADER made: the body of k_tiles_from_table (dmpnn_collate.hip), shared with
// the launch that also carries the forward's weight pre-split (k_tiles_from_table_split, dmpnn_prepare.hip).
__device__ __forceinline__ void tiles_from_table_body(const int* __restrict__ tile_row, const int* __restrict__ tile_atom,
                                                          int n_tiles, int nV, int nE, int* __restrict__ plan, PlanLayout L) {
    __shared__ int bad_s, spill_s;
    if (threadIdx.x == 0) { bad_s = 0; spill_s = 0; }
    __syncthreads();
    int* mrow = plan + L.mtile_row;
    int* matom = plan + L.mtile_atom;
    const int slots = (int)L.max_mtiles + 2;
    int bad = 0;
    for (int t = threadIdx.x; t < slots; t += blockDim.x) {
        int r = nE, a = nV;
        if (t < n_tiles) {
            r = tile_row[t]; a = tile_atom[t];
            const int r1 = t + 1 < n_tiles ? tile_row[t + 1] : nE, a1 = t + 1 < n_tiles ? tile_atom[t + 1] : nV;
            if (r < 0 || a < 0 || r1 < r || a1 < a || r1 > nE || a1 > nV) bad = 1;
            else if (a1 == a && r1 != r) bad = 1;  // edge rows in a tile without atoms: nobody would check (or compute) them
            else if (r1 - r > kMegaBM || a1 - a > kMegaBA) atomicAdd(&spill_s, 1);  // (the tile kernel's generic path; it checks closure itself)
            if (t == 0 && (r != 0 || a != 0)) bad = 1;
        }
        mrow[t] = r;
        matom[t] = a;
    }
    if (n_tiles == 0 && (nE > 0 || nV > 0)) bad = 1;
    if (bad) atomicOr(&bad_s, 1);
    __syncthreads();
    if (threadIdx.x < DMPNN_HDR_WORDS) {
        int v = 0;
        const int h = threadIdx.x;
        if (h == DMPNN_HDR_FLAGS) v = (bad_s ? PLAN_NO_PIECE_TILES : 0) | PLAN_TILES_ONLY;
        if (h == DMPNN_HDR_NMTILES) v = bad_s ? 0 : n_tiles;
        if (h == DMPNN_HDR_NSPILL) v = bad_s ? 0 : spill_s;
        if (h == DMPNN_HDR_LIGHT) v = 2;
        if (h == DMPNN_HDR_NATOMS) v = nV;
        if (h == DMPNN_HDR_NEDGES) v = nE;
        if (h == DMPNN_HDR_TILE_STRIDE) v = kFusedBM;
        plan[h] = v;
    }
}

// ... the same with the pre-split of the forward's weights riding in the launch (workgroups 1 ..: one wave per matrix row), as K0 from the
// batch vector does: the loader-tiles path pays no launch for the split either.  *did_split says whether it did.
int launch_tiles_from_table_split(const int* tile_row, const int* tile_atom, int64_t n_tiles, int64_t nV, int64_t nE, int* plan, hipStream_t s,
                                  const dmpnn_fwd_args* split_for, bool* did_split);
// mol_bounds (optional): the molecule ranges also as the table dmpnn_molagg_* read (first[n_mols] | end[n_mols] | flag)
// split_for / did_split: the pre-split of that forward's weights (tile kernel on the f16 pipe) rides in the same launch — workgroup 0
// plans, the others split; *did_split tells the caller to pass DMPNN_F_WSPLIT_READY to the forward
int launch_prepare_tiles_batch(const int64_t* edge_index, const int64_t* batch, int64_t nV, int64_t nE, int* plan, hipStream_t s,
                               int* mol_bounds = nullptr, int64_t n_mols = 0, const dmpnn_fwd_args* split_for = nullptr, bool* did_split = nullptr);
bool mega16_split_args(const dmpnn_fwd_args& a, mega16::SplitArgs* sp);
size_t fused16_h0q_bytes(const dmpnn_fwd_args& a);   // (dmpnn_step16.hip: H0 as row quads on the per-step fused route's inference forward)
bool mega16_keeps_rows(const dmpnn_fwd_args& a);   // M^(t) kept as split rows in `msplit` (the product operands of k_wgrad16r)
// dmpnn_prepare_tiles for a training step that also aggregates per molecule: *wrote_bounds says whether `mol_bounds` was filled
// (the single-workgroup planner from the batch vector does it on the side; every other planner leaves it to dmpnn_molagg_bounds)
int prepare_tiles_and_bounds(const int64_t* edge_index, const int64_t* rev, const int64_t* batch, int64_t nV, int64_t nE, void* plan,
                             size_t plan_bytes, int* mol_bounds, int64_t n_mols, void* stream, bool* wrote_bounds,
                             const dmpnn_fwd_args* split_for = nullptr, bool* did_split = nullptr);
// the same tables for batches beyond the single-workgroup plan (dmpnn_tiles_large.hip)
bool tiles_large_fits(int64_t nV, int64_t nE);
int launch_prepare_tiles_large(const int64_t* edge_index, const int64_t* batch, int64_t nV, int64_t nE, int* plan, hipStream_t s);
// batches the single-workgroup plan (and therefore the piece tiles / the whole-forward tile kernel) takes
constexpr int kSmallPlanMaxAtoms = 6144, kSmallPlanMaxEdges = 10240;
inline size_t small_plan_lds_bytes(int64_t nV, int64_t nE) {
    return ((size_t)(3 * (nV + 2)) * 4 + (size_t)(5 * nE + 2) * 2 + 31) & ~size_t(15);
}
inline bool small_plan_fits(int64_t nV, int64_t nE) {
    return nV <= kSmallPlanMaxAtoms && nE <= kSmallPlanMaxEdges && small_plan_lds_bytes(nV, nE) <= 160 * 1024 - 2048;
}
int launch_message(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                   int64_t ld_in, float* M, int64_t ld_m, int act, float slope,
                   const float* slope_ptr, unsigned flags, hipStream_t s);
int launch_aggregate(const PlanView& pv, int64_t nV, int64_t nE, int64_t h, const float* Hin,
                     int64_t ld_in, float* Mv, int64_t ld_mv, int act, float slope,
                     const float* slope_ptr, hipStream_t s);
int launch_linear(const dmpnn_gemm_args& a, hipStream_t s);

// Internal extensions of a contraction launch (not part of the C ABI): second gather, row tiles
// from the plan, the fused segment epilogue, NaN poisoning driven by the plan flags.
struct GemmExtra {
    const int* gather2; int64_t gather2_rows;   // row gather of A2 (E[perm] in the fused initialize)
    const int* tile_row; int n_tiles;           // row tiles (plan tile table); null = uniform 16*RT-row tiles
    bool seg;                                   // EPI_SEG: rows are CSR-ordered edges, tiles hold whole atoms
    const int* tile_atom; const int* row_ptr; const int* revp;
    float* Mout; int64_t ldm;                   // Mout[revp[r]] = S[dst(r)] - Y[r]
    float* Sout; int64_t lds;                   // Sout[v] = sum of Y over the rows of v
    const int* poison_flags; int poison_mask;   // plan header word 0 and the mask that makes the output NaN
};
int launch_linear_ex(const dmpnn_gemm_args& a, const GemmExtra& x, hipStream_t s);
// whole-forward tile kernel (dmpnn_mega.hip): writes tau(W_o[V || Mv] + b_o) to out[ldout]
bool mega_shapes_ok(const dmpnn_fwd_args& a);
int launch_mega_forward(const dmpnn_fwd_args& a, float* out, int64_t ldout, hipStream_t s);
// the same on the f16 matrix pipe with the exact 3-term split (dmpnn_mega16.hip)
size_t mega16_wsplit_bytes(const dmpnn_fwd_args& a);
size_t mega16_fwd_wsplit_bytes(const dmpnn_fwd_args& a);
// per-step split-MFMA contraction (dmpnn_rows16.hip): pre-split weights of one matrix, shape gate, launch
struct SplitWView { const unsigned char* p; const float* inv_scale; int nc; };
size_t linear16_wsplit_bytes(int64_t N, int64_t K);
int split_weights_view(const float* W, int64_t ldw, int64_t N, int64_t K, int tr, void* ws, SplitWView* out, hipStream_t s);
SplitWView split_weights_view_of(void* ws, int64_t N, int64_t K);  // descriptor of a workspace that already holds the pre-split
struct SplitWJob { const float* W; int64_t ldw, N, K; int tr; void* ws; };
int split_weights_views(const SplitWJob* jobs, int n, SplitWView* out, hipStream_t s);   // up to 6 matrices, ONE launch
bool split_weights_args(const SplitWJob* jobs, int n, SplitWView* out, mega16::SplitArgs* sp);   // ... the argument block only
int launch_split_args(const mega16::SplitArgs& sp, hipStream_t s);
bool linear16_ok(const dmpnn_gemm_args& a);
int launch_linear16_view(const dmpnn_gemm_args& a, const SplitWView& W, const int* poison_flags, int poison_mask, hipStream_t s);
int launch_mega16_forward(const dmpnn_fwd_args& a, float* out, int64_t ldout, hipStream_t s);
// waves per tile workgroup of the whole-forward / backward tile kernels for this batch (4, or 8 when the launch has at most one tile per CU)
int tile_waves(const dmpnn_fwd_args& a, int n_tiles);
// per-step fused route on the f16 pipe (dmpnn_step16.hip): inference forward, any molecule size, d_h <= 320
bool fused16_shapes_ok(const dmpnn_fwd_args& a);
// its LEAN training forward (dmpnn_step16.hip) and the backward that reads what it keeps (dmpnn_bstep16.hip)
bool fused16_lean_shapes(const dmpnn_fwd_args& a);     // shapes / options only (the route rule)
size_t fused16_lean_bits_bytes(const dmpnn_fwd_args& a);
bool fused16_lean(const dmpnn_fwd_args& a);            // ... and the workspace is there (DMPNN_F_KEEP, keep_bits, msplit, H0)
int64_t split_row_floats(int64_t d_h);

// ---- weight gradients on the f16 pipe (dmpnn_wgrad16.hip): operands split once into transposed blocks, then the products ----
struct WSplitJob {
    int64_t M; int C;                                            // reduction rows; logical columns (incl. the column of ones)
    const float* A1; int64_t lda1; const int* g1; int K1;        // columns [0, K1): A1[g1(m)] (g1 null: rows in place)
    const float* A2; int64_t lda2; const int* g2; int K2;        // columns [K1, K1 + K2): A2[g2(m)]
    int ones;                                                    // column K1 + K2 is 1 (the bias gradient = column sums)
    unsigned char* out; float* scales;                           // [column tile][chunk][8 KB]; [column tile][chunk]
    int n_ct, n_chunks, wg0;
    const long long* g1_64; int64_t g1_rows;                     // A1's row gather as int64 (the caller's own src array under a tile plan) when g1
                                                                 // is null: unvalidated, so clamped into [0, g1_rows)
};
struct WSplitArgs { WSplitJob job[8]; int n_jobs; };
struct WProdArgs {
    const unsigned char* Z; const float* sZ; const unsigned char* A; const float* sA;
    int n_nt, n_kt, n_chunks, chunks_per_split, splits;
    int N, Kt; float* slab; int ldk; int64_t slab_stride;
};
struct WProdPlan { int n_nt, n_kt, n_chunks, chunks_per_split, splits, ldk; int64_t slab_stride; };
constexpr int kWProdMaxJobs = 8;
struct WProdJobs { WProdArgs job[4]; int wg0[5]; int n_jobs; };  // several products in one launch: job j owns workgroups [wg0[j], wg0[j + 1]), multiples of 8
// products over operands in SPLIT-ROW form (rows of [hi 32 | lo 32] chunks + a tail with the row's scale: k_wgrad16r, round 4) — the
// operands are consumed as the step kernels keep them, nothing is re-blocked
struct WProdRPlan { int n_kg, splits, rows_per_split, ldk; int64_t slab_stride; };
// rows_per_split 0: the round-4 rule (~256 workgroups per JOB); else the launch's common row count per workgroup (wgrad16r_rows_per_split)
WProdRPlan plan_wgrad16r(int64_t M, int N, int K, int rows_per_split = 0);   // (K = 1: the plan of a column-sum job)
// Round 6: the row counts per workgroup of ALL jobs of one launch (reduction rows M[j], K[j] columns of A; K <= 1: a column-sum job):
// rows_out[j] in inverse proportion to the job's cost per 32-row stage, sized so that the launch is one workgroup per CU (less `reserve`
// for a rider's jobs), or whole rounds of >= 512-row workgroups for long reductions.  The slabs the products write and the reduce
// launch reads back are splits x N x K floats per job: at ~256 workgroups per JOB (round 4) that was 84 MB per 512-molecule step
// against 50 MB of operands.
void wgrad16r_plan_launch(const int64_t* M, const int* K, int n, int reserve, int* rows_out);
constexpr int kWProdRMaxJobs = 16;
// one job: gW[N][K] = Z^T A over M rows into `plan.splits` slabs [N][plan.ldk] at `slab`;  A == nullptr: the column sums of Z (a bias
// gradient) into column 0 of slabs [N][4]
// slab_b (or null; products only): the column sums of Z as well, into column 0 of `plan.splits` slabs [N][4] — they ride in the product's workgroups
struct WProdRJob { const unsigned char* Z; int tsz; const unsigned char* A; int tsa; int64_t M; int N, K; float* slab; WProdRPlan plan; float* slab_b; };
int launch_wgrad16r(const WProdRJob* jobs, int n, hipStream_t s);
// fp32 rows [A1[g1] || A2[g2]] -> split rows [M][ts] (k_rows2sr): the product operands no kernel already holds split
struct SRJob { const float* A1; int64_t lda1; int K1; const int* g1; const long long* g1_64; int64_t g1_rows;
               const float* A2; int64_t lda2; int K2; const int* g2; int64_t M; unsigned char* out; int ts; };
int launch_rows2sr(const SRJob* jobs, int n, hipStream_t s);
size_t wsplit16_bytes(int64_t M, int64_t C);
// One more weight-gradient product for the launches of a backward pass on the f16 pipe (the predictor's first layer in a training
// step: gW[N, K] = Z^T A, gb = colsum(Z) — 512 rows are 16 chunks beside the block's 1 140): rides in k_wsplit16 / k_wgrad16 /
// k_wgrad_reduce_multi instead of three launches of its own.  ws: >= extra_wgrad_ws_floats(M, N, K + ones) floats, 16-byte aligned.
// Round 5: what dmpnn_train_step hands the tile kernels' training forward so that it leaves the per-molecule aggregate of its output
// (Mega16K::agg_*): set on the calling thread around its dmpnn_forward call, taken (and cleared) by the tile kernel's launcher.
struct AggRide {
    float* Hm; int ld; const int64_t* batch; int* table; int n_mols; int mode; float norm;   // table: dmpnn_molagg's first | end | flag | done
    bool taken;
};
extern thread_local AggRide g_agg_ride;
struct ExtraWgrad {
    const float* Z; int64_t ldz; const float* A; int64_t lda; int64_t M; int N, K, ones;
    float* gW; int64_t ldgw; float* gb; float* ws;
};
size_t extra_wgrad_ws_floats(int64_t M, int N, int Kt);
int backward_impl(const dmpnn_bwd_args* b, void* stream, const ExtraWgrad* extra, bool* extra_done);
void wsplit16_job(WSplitJob* j, int64_t M, int C, const float* A1, int64_t lda1, const int* g1, int K1, const float* A2, int64_t lda2,
                  const int* g2, int K2, int ones, void* ws);
int launch_wsplit16(WSplitArgs& a, hipStream_t s);
bool wgrad16_operand_ok(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2);
WProdPlan plan_wgrad16(int64_t M, int N, int Kt);
void wgrad16_add(WProdJobs* jobs, const WSplitJob& Z, const WSplitJob& A, const WProdPlan& p, int N, int Kt, float* slab);
int launch_wgrad16(const WProdJobs& jobs, hipStream_t s);
// pending (or null): the pre-split of the weights, not launched yet — it rides in the forward's first launch when that launch does
// not read the weights (k_split_rows), else it is launched first
int launch_fused16_forward(const dmpnn_fwd_args& a, const SplitWView* w16, float* out, int64_t ldout, hipStream_t s,
                           const mega16::SplitArgs* pending = nullptr);
// the backward step kernels of that route's lean training forward (dmpnn_bstep16.hip)
// (Zrows: gZ of the site as split rows [n_edges][split_row_bytes(d_h)], a product operand of k_wgrad16r)
int launch_bstep16(const dmpnn_fwd_args& f, int site, const float* Tin, const float* gMv, const SplitWView* W, float* Tout,
                   unsigned char* Zrows, hipStream_t s);
// the data-gradient chain of the backward pass as one tile kernel (dmpnn_mega16_bwd.hip)
size_t mega16_bwd_wsplit_bytes(int64_t h);
// (rows: gZ^(t) [depth - 1 slots] / gH0 / gZO as split rows of tsr bytes instead of the fp32 tensors — see Mega16BwdK)
struct Mega16BwdRows { unsigned char* gZ; unsigned char* gH0; unsigned char* gZO; int tsr; };
int launch_mega16_backward(const dmpnn_fwd_args& f, const float* gHO, int64_t ldg, const float* HO, int64_t ldho, float* gZO,
                           float* gZs, float* gH0, void* wsplit, float* sp_gM, float* sp_Ta, hipStream_t s, const float* g_edge = nullptr, int64_t ld_gedge = 0,
                           const Mega16BwdRows* rows = nullptr);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace dmpnn
