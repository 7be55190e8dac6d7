// Whole BondMessagePassing.forward for one tile of whole molecules in one launch — fp32-equivalent
// arithmetic on the f16 matrix pipe ("3 x f16 split").
//
// gfx950 has no reduced-precision fp32 matrix path (no xf32/TF32): the exact fp32 MFMA runs at the
// vector rate (157 TF), 1/16 of the f16 MFMA rate.  Every fp32 operand x is therefore split exactly,
// after an exact power-of-two scaling s that puts the operand's largest magnitude at 2^13..2^14,
//     x * s = hi + lo + e,   hi = f16(x s),  lo = f16(x s - hi),  |e| <= 2^-24 |x s|  (or <= 2^-25 absolute)
// and a product of two fp32 numbers becomes three f16 MFMA products with fp32 accumulation,
//     a b  ~  hi_a hi_b + hi_a lo_b + lo_a hi_b        (f16 x f16 is exact in fp32; the dropped terms are
//                                                        <= 3 * 2^-24 |a b|: fp32 rounding class)
// 3 x v_mfma_f32_16x16x32_f16 instead of 8 x v_mfma_f32_16x16x4_f32 per 16x16x32 block: 5.3x the
// matrix throughput at the accuracy of the fp32 path (oracle parity 1e-7-class, tests/).
// Scales: one per tile and contraction for the A operand (tile maximum, LDS atomic), one per output
// column for the weights (pre-split once per forward by k_split_weights into the staging layout
// [N][chunk][hi 32 | lo 32] halfs, so the B staging is a plain 16-byte copy without tail handling).
//
// Structure, tiling, LDS ring, piece tiles and the segment epilogue are those of dmpnn_mega_impl.hpp;
// the fp32 epilogue tile overlays the (idle) B ring, the split A tile of the next step has its own region.
#pragma once

#include <type_traits>

#include "dmpnn_mega_impl.hpp"

namespace dmpnn {
namespace mega16 {

using gemm::BK;
using gemm::f32x4;
using gemm::kOOB;
using gemm::kThreads;
using gemm::rsrc_t;
using gemm::u32x4;
using gemm::u32x2;
using mega::RT_A;
using mega::RT_E;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));


// Packed pre-split weight matrix in fragment-major order [column tile][chunk][hi|lo][lane][16 B] (see k_split_weights),
// plus the inverse scale of every row.
struct SplitW {
    const unsigned char* p;  // [ceil(N/16)][nc][2][64][16 B]
    const float* inv_scale;  // [N]  1 / s_n
    int nc;
};

struct Mega16K {
    mega::MegaK m;           // same graph / feature / output description as the fp32 kernel
    SplitW Wi, Wh, WoM, WoV; // W_i [N, d_v + d_e], W_h [N, N], W_o[:, d_v:], W_o[:, :d_v]  (WoM and WoV share inv_scale)
    // DMPNN_F_ATOM (atom messages, inference): W_i is [N, d_v], Wh holds W_h[:, :N] and WhE the bond-feature block W_h[:, N:N + d_e]
    // (one chunk: d_e <= 32); atom_de = d_e (the kernel's own d_e is 0: E is not part of the K1 operand), 0 = the bond variant
    SplitW WhE; int atom_de;
    // ... training (DMPNN_F_ATOM | DMPNN_F_KEEP): the bond-feature half of the message, ME[r'] = (sum E)[src r'] as [n_edges][16] fp32 rows
    // (columns >= d_e zero), written once per kept message slot (me_slot floats apart) — dmpnn_backward's W_h product then reads
    // [M^(t) || ME] as ONE operand over all steps' rows
    float* atom_me; long long me_slot;
    // Round 4 (bond messages, training): M^(t) is kept as SPLIT ROWS — the very pieces the next contraction's A tile is made of, written
    // to memory beside it with the tile's scale in the row tails (rows of `tsr` bytes, slot t - 1 at mrow_slot bytes) — because that is
    // what the weight-gradient product reads (k_wgrad16r): no fp32 copy, no re-blocking pass.  null: fp32 rows in m.Ms as before
    // (atom messages; the generic path of a molecule beyond the tile still writes m.Ms and converts its rows at the end)
    unsigned char* Mrows; int tsr; long long mrow_slot;
    // dmpnn_fwd_args.keep_bits (training on a tile plan, ReLU-class activation, no dropout): H0 and H^(t) leave the kernel as ONE
    // bit per element — [x > 0], all the backward tile kernel needs of them — in the order of the matrix-pipe fragments: word
    // (rt WN + ct) 4 + r of wave w of tile t (64 words per wave, 256 per tile) is the ballot of "element (row rt 16 + 4 lg + r, column
    // 16 (w WN + ct) + li) > 0" over the wave's lanes.  Slot 0 = H0, slot t = H^(t); bits_slot = words per slot.
    unsigned long long* keep_bits; long long bits_slot;
    // Round 5 (a whole training step, dmpnn_train_step): the per-molecule aggregate of the finalize output straight from the tile — a tile
    // holds whole molecules, their H_v rows are in LDS when the kernel ends.  agg_Hm [n_mols][agg_ld] receives agg(H_v) of every molecule
    // of a regular tile (rows added in increasing atom order: dmpnn_molagg_fwd's arithmetic) and agg_done[m] = 1; a molecule the tile
    // kernel does not carry to its end (a spill tile, an unused slot) keeps done = 0 and is aggregated by the head's column kernel from
    // H_v as before.  agg_bounds: first[n_mols] | end[n_mols] (the table K0 wrote); null agg_Hm: off.
    float* agg_Hm; int agg_ld; const long long* agg_batch; const int* agg_bounds; int* agg_done; int agg_n_mols; int agg_mode; float agg_norm;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// exact power-of-two scale that puts `maxabs` at [2^13, 2^14); 1 for 0 / inf / nan
__device__ __forceinline__ float scale_for(float maxabs) {
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.f;
    int e;
    frexpf(maxabs, &e);  // maxabs = m 2^e, m in [0.5, 1)
    // (the exponent is kept within +-126: the reciprocal of every scale — one integer subtraction on the exponent, rcp_pow2 in
    //  dmpnn_step16_impl.hpp — must be a normal number too; an operand of magnitude < 2^-112 is then split with less headroom, never to 0)
    const int k = 14 - e;
    return ldexpf(1.f, k > 126 ? 126 : (k < -126 ? -126 : k));
}
// store of a kept / gradient tensor row piece (16 bytes).  DMPNN_NT_KEEP (experiment): non-temporal — the tensors a training step
// streams out are read again only by a later launch
__device__ __forceinline__ void store_keep4(float* p, float4 v) {
#if defined(DMPNN_NT_KEEP)
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
// The exact pair split of two elements, x s = hi + lo: hipcc emits v_pk_mul_f32, v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_fma_f32,
// v_cvt_pk_f16_f32 per pair (3 VALU instructions per element).  Round 6 measured the 2-per-element form on the mixed-precision FMA
// (v_fma_mixlo/hi_f16 hi, x, s, -0;  v_fma_mixlo/hi_f16 lo, x, s, -hi.f16 — bit for bit the same words, scripts/micro/split_probe.hip):
// the tile kernel's chain 65.3 k -> 64.5 k cycles, configs 2-4 and the training step unchanged — the phases between the contractions
// are dependency chains, not VALU issue — and v_fma_mixhi_f16 is a partial register write whose consumer needs a wait state the
// compiler cannot place for inline assembly (gfx940 dst-sel forwarding hazard: wrong results when the consumer followed directly).
// Not kept (profiles/r06_split_on_fma_mix.txt).
__device__ __forceinline__ void split2(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    const float a = x0 * s, b = x1 * s;
    const h2 h = h2{(_Float16)a, (_Float16)b};
    const h2 l = h2{(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split4(float4 x, float s, h4& hi, h4& lo) {
    unsigned h0, l0, h1, l1;
    split2(x.x, x.y, s, h0, l0);
    split2(x.z, x.w, s, h1, l1);
    hi = __builtin_bit_cast(h4, u32x2{h0, h1});
    lo = __builtin_bit_cast(h4, u32x2{l0, l1});
}
// 1 / s for a power of two s in [2^-126, 2^126] (every scale of the split format is one: scale_for, k_split_weights) — EXACT, one integer
// subtraction instead of the ~10 VALU instructions of an IEEE division
__device__ __forceinline__ float rcp_pow2_exact(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }

// ---- pre-split of the weights: one wave per (matrix, row) -----------------------------------------
struct SplitJob {
    const float* W; int ldw; int col0; int K;   // source columns [col0, col0 + K) of row n
    int scale_col0, scale_K;                    // columns the row scale is taken over (W_o: the whole row)
    unsigned char* out; int nc; float* inv_scale;
    int tr;                                     // 1: the matrix is given transposed, element (n, k) = W[k * ldw + n]
    int N;                                      // rows of THIS matrix (0: SplitArgs.N — jobs of one launch may differ in height)
};
struct SplitArgs {
    SplitJob job[8];   // (forward: 4; + the backward tile kernel's 2 for a training forward; + 1 for atom messages)
    int n_jobs, N;
};
// one wave per (matrix, row).  A device function: it also rides in the launch of K0 (dmpnn_prepare.hip: workgroup 0 builds the
// tile table, the others split the weights — the split then costs no launch and no time of its own, so NOTHING about the
// weights is cached between forwards)
__device__ __forceinline__ void split_weights_wave(const SplitArgs& a, int wave, int lane) {
    const int Np = (a.N + 15) & ~15;  // whole column tiles: the padding rows of the last tile are written as zeros
    const int j = wave / Np, n = wave - j * Np;
    if (j >= a.n_jobs) return;
    const SplitJob& J = a.job[j];
    const int NJ = J.N > 0 ? J.N : a.N;
    if (n >= ((NJ + 15) & ~15)) return;   // (a shorter matrix in a launch sized for the tallest)
    const bool live = n < NJ;
    const long long rs = J.tr ? 1 : J.ldw, ks = J.tr ? J.ldw : 1;  // strides of the output index n and of the reduction index k
    const float* row = J.W + (long long)(live ? n : 0) * rs;
    float mx = 0.f;
    for (int k = lane; k < J.scale_K; k += 64) mx = fmaxf(mx, fabsf(row[(J.scale_col0 + k) * ks]));
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float s = live ? scale_for(mx) : 0.f;
    if (lane == 0 && J.inv_scale && live) J.inv_scale[n] = 1.f / s;
    // Fragment-major layout: the 16 B a lane of the tile kernel loads for (column tile T = n / 16, chunk c,
    // part hi|lo) are contiguous per wave instruction: [T][c][part][lg][li][8 halfs]  (1 KiB per instruction).
    _Float16* out = reinterpret_cast<_Float16*>(J.out);
    const int T = n >> 4, li = n & 15;
    for (int k = lane; k < J.nc * 32; k += 64) {
        const float x = (k < J.K && live) ? row[(J.col0 + k) * ks] * s : 0.f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        const int c = k >> 5, kk = k & 31, lg = kk >> 3, j = kk & 7;
        const long long base = (((long long)T * J.nc + c) * 2) * 512 + (lg * 16 + li) * 8 + j;
        out[base] = hi;
        out[base + 512] = lo;
    }
}
static __global__ __launch_bounds__(256) void k_split_weights(SplitArgs a) {
    split_weights_wave(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// fp32 rows -> split rows (one wave per row, its own power-of-two scale): the generic path's kept tensors, for a molecule beyond the tile
__device__ __forceinline__ void rows_to_sr(const float* src, long long ld, long long row0, int nrows, int N, unsigned char* dst, int ts,
                                           int wave, int lane, int n_waves) {
    for (int r = wave; r < nrows; r += n_waves) {
        const float* x = src + (row0 + r) * ld;
        float v[8];
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < N ? x[c] : 0.f;
            mx = fmaxf(mx, fabsf(v[j]));
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float sc = scale_for(mx);
        unsigned char* o = dst + (row0 + r) * ts;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane + 64 * j;
            if (c < N) {
                const float y = v[j] * sc;
                const _Float16 hi = (_Float16)y;
                *reinterpret_cast<_Float16*>(o + (c >> 5) * 128 + (c & 31) * 2) = hi;
                *reinterpret_cast<_Float16*>(o + (c >> 5) * 128 + 64 + (c & 31) * 2) = (_Float16)(y - (float)hi);
            }
        }
        if (lane == 0) *reinterpret_cast<float4*>(o + (ts - 16)) = make_float4(sc, mx > 0.f ? 0.f : 1.f, 0.f, 0.f);   // (scale, zero-row flag: k_wgrad16r)
    }
}

// LDS: ONE tile region holds, at different times, the split A tile of a contraction, the K1 / V operand staging tile and
// the fp32 tile of a row-major store (the barriers that separate those uses are in the kernel) + metadata + incidence
// fragments: 73 KB at d_h = 300, so TWO workgroups fit a CU.  With the register budget of two waves per SIMD
// (__launch_bounds__(256, 2)) consecutive tiles on one CU overlap their latency-bound phases: measured on MI355X
// 41.1 -> 35.4 us at 512 molecules (one tile per CU: fewer registers, a compact contraction loop) and 301.6 -> 183.6 us at
// 4 096 molecules (1 839 tiles, 7.2 per CU).
#if !defined(DMPNN_TILE8_BRING)
#define DMPNN_TILE8_BRING 1   // weight-fragment sets of the 8-wave form's contraction: 1 = the one-set ring; 2 (chunk c + 2 behind chunk c) measured 0.7 us SLOWER at 512 molecules (profiles/r06_tile8_first_ab.txt: the stream is throughput-, not latency-bound)
#endif
template <int WN>
constexpr size_t tile_region_bytes() {
    constexpr size_t ts = 64 * WN * 4 + 16, tsg = 4 * 128 + 16;  // split A tile row | staging tile row (the larger one for d_h <= 64)
    return (size_t)kMegaBM * (ts > tsg ? ts : tsg);
}
constexpr int kAtomK = 16;  // atom variant: bond features per row it stages (d_e <= 16: the v2 featurizer has 14)
constexpr size_t kAtomLds = (size_t)(kMegaBM + kMegaBA) * kAtomK * sizeof(float);  // E rows of the tile [48][16] | per-atom sums [32][16]: 5 KB
template <int WN>
constexpr size_t lds_bytes() {
    return tile_region_bytes<WN>() + (size_t)(3 * kMegaBM + kMegaBA + 24) * sizeof(int) + 10 * 64 * 16;
}
template <int WN>
constexpr size_t lds_bytes_atom() { return lds_bytes<WN>() + kAtomLds; }

// SA (simple activation): identity / ReLU / LeakyReLU / PReLU are a compare + select in line; tanh and ELU get their own
// instantiation — a workgroup runs its code ONCE, cold: every KB of inlined transcendental code that the ReLU path has to
// jump across costs instruction fetches (the kernel was 203 KB against a 64 KB instruction cache).
// KEEP: the training forward (H0, H^(t), M^(t), Mv stream out for the backward pass); the inference instantiation carries none of
// that code (64-bit row addresses and a divergent branch per stored fragment).
//
// NW (round 6): waves per workgroup.  4 = the form above (wave w owns column tiles WN w .. WN w + WN - 1; two workgroups per CU when
// the launch has more tiles than CUs).  8 (d_h in (128, 320] only: 20 column tiles) = ONE tile as a 512-thread workgroup, the column
// tiles split 3+3+3+3+2+2+2+2: waves w and w + 4 share a SIMD, so every SIMD still carries 5 column tiles, but as TWO waves whose
// dependent chains are 3/5 and 2/5 as long and cover each other's waits (weight fragments, LDS, the epilogues' conversion chains).
// Picked when the launch has at most one tile per CU (launch_mega16_forward): there the 4-wave form leaves every SIMD with ONE wave.
// Each wave class runs its own instantiation of the tile's body (exact vmcnt waits, no guards in the MFMA loops): together they
// are as much code as the 4-wave body.
// LP (round 6, DMPNN_F_STORE16 on this route: OPT-IN, NOT fp32-class): every matrix product on the HI halves alone — operand rows, messages,
// weights and the H of the incidence products as ONE f16 per element (11-bit significands under the same power-of-two scales; bf16, which
// BASELINE configs[1] names, has 8), one MFMA pass instead of three, half the weight stream; accumulation, residual, bias, activation and the
// output stay fp32.  Inference only (KEEP = false), bond messages.
template <int WN, bool SA, bool KEEP, int NW = 4, bool LP = false>
__global__ __launch_bounds__(64 * NW, 2) void k_mpnn_tile16(Mega16K G) {
    static_assert(!(LP && KEEP), "the hi-halves-alone form is inference only");
    static_assert(NW == 4 || (NW == 8 && WN == 5), "the 8-wave form splits 20 column tiles 3+3+3+3+2+2+2+2");
    const mega::MegaK& g = G.m;
    constexpr int KT = 64 * NW;                // threads of the workgroup
    constexpr int JR = 16 / NW;                // gathered operand rows per thread and row tile: row = wave + NW j
    constexpr int BM = kMegaBM, BA = kMegaBA, BN = 64 * WN, LDC = BN + 4, QN = BN / 4;
    constexpr int TS = BN * 4 + 16;            // bytes of one row of the split A tile: BN/32 chunks x 128 + 16
    constexpr int ITEMS = (BM * QN + KT - 1) / KT;   // (row, quad) items per thread of a row-major tile pass (guarded by r < n_r)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* T16 = lds;                               // [BM][TS]   split A operand of the LDS-A contractions
    static_assert(LDC * 4 == TS, "the fp32 tile and the split A tile have the same footprint");
    constexpr int META_OFF = (int)tile_region_bytes<WN>();
    float* T = reinterpret_cast<float*>(lds);               // [BM][LDC] fp32 tile of the row-major stores: overlays T16 / Ag
    int* revl = reinterpret_cast<int*>(lds + META_OFF);     // [BM]
    int* aor = revl + BM;                                   // [BM]
    int* asrc = aor + BM;                                // [BM] tile-local source atom of a row (tile plan only)
    int* rp = asrc + BM;                                     // [BA + 1]
    unsigned* maxbits = reinterpret_cast<unsigned*>(rp + BA + 1);  // [0..3] tile maxima (float bits, rotating), [5] tile-not-closed flag
    // [10][64] incidence fragments of the segment MFMAs (see segment_mfma): 16-byte aligned behind the metadata
    h8* cfrag = reinterpret_cast<h8*>(lds + META_OFF + (((3 * BM + BA + 1 + 8) * 4 + 15) / 16) * 16);

    using T_ = std::true_type;
    using F_ = std::false_type;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int li = lane & 15, lg = lane >> 4;
    int kq = tid & 7;
    auto launder = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4; kq = tid & 7;
    };
#if !defined(DMPNN_NO_KERNARG_WARM)
    warm_kernargs<(int)sizeof(Mega16K)>();
#endif
    int n_stamp = 0;
    auto stamp = [&]() {
        if (g.dbg && blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 32) g.dbg[n_stamp] = (long long)__builtin_readcyclecounter();
        ++n_stamp;
    };
    stamp();  // 0: kernel entry
    const int t = blockIdx.x;
    // The tile's four table words and the plan's two header words in ONE round trip: hipcc sinks the table loads below the poison
    // branch otherwise (they are dead on that path), which made the header a dependent round trip of its own in front of them —
    // four round trips before the first operand byte is requested instead of three (ISA, round 6).
    int rs = g.mtile_row[t], re = g.mtile_row[t + 1];
    int va = g.mtile_atom[t], vb = g.mtile_atom[t + 1];
    int hdr_light = g.flags[DMPNN_HDR_LIGHT], hdr_flags = g.flags[0], hdr_tiles = g.flags[DMPNN_HDR_NMTILES];   // (flags = the header's address)
    asm volatile("" : "+v"(rs), "+v"(re), "+v"(va), "+v"(vb), "+v"(hdr_light), "+v"(hdr_flags), "+v"(hdr_tiles));
    rs = __builtin_amdgcn_readfirstlane(rs); re = __builtin_amdgcn_readfirstlane(re);
    va = __builtin_amdgcn_readfirstlane(va); vb = __builtin_amdgcn_readfirstlane(vb);
    hdr_light = __builtin_amdgcn_readfirstlane(hdr_light); hdr_flags = __builtin_amdgcn_readfirstlane(hdr_flags);
    hdr_tiles = __builtin_amdgcn_readfirstlane(hdr_tiles);   // (tiles the plan really has: the launch's grid is only a bound)
    const int nrows = re - rs, na = vb - va;
    const int N = g.h, qn = N >> 2;
    // Tile plan (header LIGHT == 2): the rows of a tile are its edges in the CALLER's order, src / dst / rev come
    // straight from the caller's int64 arrays and the tile checks by itself that it is closed (every edge id in
    // its range has both atoms and its reverse edge inside the tile; by counting, no other edge then enters its
    // atoms).  A tile that is not closed writes NaN to its atoms.
    const bool lean = hdr_light == 2;
    // (a training forward on a tile plan keeps its tensors in the caller's edge order: dmpnn_backward takes them so when told
    //  DMPNN_F_TILE_PLAN, dmpnn_mega16_bwd_impl.hpp)
    // (round 6: a caller that knows an upper bound of the tile count — the batch's molecule count: a tile holds at least one — launches only
    //  that many workgroups instead of the layout's bound, dmpnn_fwd_args.n_tiles_launch.  A plan with MORE tiles than the launch has
    //  workgroups means that bound was wrong: every output NaN, never a batch with rows nobody computed)
    // (measured and not kept, profiles/r06_tile_loop_not_kept.txt: the kernel's body as a loop over tiles b, b + grid, ... so that the grid can be
    //  an ESTIMATE of the tile count — 277 instead of 512 workgroups.  Correct on every grid down to one workgroup, but the back edge cost the
    //  allocation 1 .. 7 registers and the schedule 0.8 us at 64 molecules, 2 us at 512: more than the 0.5 us the idle workgroups take)
    const bool poison = (hdr_flags & (lean ? kPlanNoMegaLean : g.poison_mask)) != 0 || (lean && g.nE > 0 && (!g.edge_index || !g.rev64)) ||
                        hdr_tiles > (int)gridDim.x;
    if (poison) {
        const float nanv = __int_as_float(0x7fc00000);
        const long long total = (long long)g.nV * N;
        for (long long i = (long long)blockIdx.x * KT + tid; i < total; i += (long long)gridDim.x * KT)
            g.out[(i / N) * g.ldout + (i % N)] = nanv;
        return;
    }
    if (na <= 0 || nrows < 0 || va < 0 || vb > g.nV || rs < 0 || re > g.nE) return;  // (an unused tile slot)
    if (nrows > BM || na > BA) {
        if (NW > 4 && threadIdx.x >= kThreads) return;  // (the generic path is written for 256 threads; s_barrier counts live waves only)
        // a piece (molecule) larger than the matrix-pipe tile — the reference has no size limit (data/collate.py:48-56):
        // the generic fp32 path carries it, whatever its size (dmpnn_spill_impl.hpp)
        const mega::MegaK& gs = spill::fresh_kernargs<Mega16K>()->m;
        if ((KEEP && gs.drop_thr) || G.atom_de) {  // the generic path has neither dropout nor atom messages: such a molecule is LOUD (NaN), never wrong
            const float nanv = __int_as_float(0x7fc00000);
            for (int i = tid; i < na * N; i += kThreads) gs.out[(long long)(va + i / N) * gs.ldout + (i % N)] = nanv;
            if constexpr (KEEP) {  // ... and so are its kept edge states, which a second read-out may consume (mab.py)
                if (gs.Hs)
                    for (int i = tid; i < nrows * N; i += kThreads)
                        for (int sl = 0; sl < gs.depth - 1; ++sl) gs.Hs[(long long)sl * gs.slot + (long long)(rs + i / N) * gs.ldh + (i % N)] = nanv;
            }
            return;
        }
        spill::forward(mega::spill_view(gs, gs.flags[DMPNN_HDR_LIGHT] == 2, rs, nrows, va, na, gs.slope_ptr ? *gs.slope_ptr : gs.slope),
                       reinterpret_cast<float*>(lds));
        if constexpr (KEEP) {
            const Mega16K& Gs = *spill::fresh_kernargs<Mega16K>();
            if (Gs.Mrows && gs.Ms) {   // the kept messages of this molecule, fp32 rows in Ms, also as the split rows the product reads
                __threadfence_block();
                __syncthreads();
                for (int sl = 0; sl < gs.depth - 1; ++sl)
                    rows_to_sr(gs.Ms + (long long)sl * gs.slot, gs.ldh, rs, nrows, gs.h, Gs.Mrows + (long long)sl * Gs.mrow_slot, Gs.tsr,
                               (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), kThreads / 64);
            }
        }
        return;
    }
    const float slope = g.slope_ptr ? *g.slope_ptr : g.slope;
    const float neg_slope = g.act == DMPNN_ACT_NONE ? 1.f : (g.act == DMPNN_ACT_RELU ? 0.f : slope);
    constexpr bool simple_act = SA;

    // index loads first: the tile metadata and the gather rows of the K1 operand (row wave + NW j of the tile)
    int revl_v = 0, rp_v = 0, aor_v = 0, asrc_v = 0;
    bool row_bad = false;
    unsigned ro1[JR * RT_E], ro2[JR * RT_E];
    {
        int i1[JR * RT_E], i2[JR * RT_E];
        if (lean && nrows > 0) {
            const int e = rs + (tid < nrows ? tid : 0);
            const long long es = g.edge_index[e], ed = g.edge_index[(long long)g.nE + e], er = g.rev64[e];
            const long long r_l = er - rs, d_l = ed - va, s_l = es - va;
            row_bad = tid < nrows && (r_l < 0 || r_l >= nrows || d_l < 0 || d_l >= na || s_l < 0 || s_l >= na);
            revl_v = (tid < nrows && !row_bad) ? (int)r_l : 0;
            aor_v = (tid < nrows && !row_bad) ? (int)d_l : 0;
            asrc_v = (tid < nrows && !row_bad) ? (int)s_l : 0;
#pragma unroll
            for (int j = 0; j < JR * RT_E; ++j) {
                const int r = wave + NW * j;
                const long long sj = g.edge_index[r < nrows ? rs + r : rs];
                i1[j] = (sj >= 0 && sj < g.nV) ? (int)sj : 0;
                i2[j] = rs + r;
            }
        } else if (lean) {  // a tile of single atoms: no rows, nothing to read
#pragma unroll
            for (int j = 0; j < JR * RT_E; ++j) { i1[j] = 0; i2[j] = 0; }
        } else {
            revl_v = tid < nrows ? g.revp[rs + tid] - rs : 0;
            rp_v = g.row_ptr[va + (tid <= na ? tid : na)] - rs;
#pragma unroll
            for (int j = 0; j < JR * RT_E; ++j) {
                const int r = wave + NW * j;
                i1[j] = g.srcp[r < nrows ? rs + r : 0];
                i2[j] = g.perm[r < nrows ? rs + r : 0];
            }
        }
#pragma unroll
        for (int j = 0; j < JR * RT_E; ++j) {
            const bool ok = wave + NW * j < nrows;
            ro1[j] = ok ? (unsigned)i1[j] * (unsigned)g.ldv * 4u : kOOB;
            ro2[j] = ok ? (unsigned)i2[j] * (unsigned)g.lde * 4u : kOOB;
        }
    }
    // the first 128 columns of [V[src] | E] are in flight while the LDS metadata is built
    const rsrc_t rVg = gemm::make_rsrc(g.V, g.v_bytes), rEg = gemm::make_rsrc(g.E, g.e_bytes);
    u32x2 a_grp[JR * RT_E];
    {
        const int k = lane * 2;
        const unsigned k1o = k < g.d_v ? (unsigned)k * 4u : kOOB;
        const unsigned k2o = (k >= g.d_v && k < g.d_v + g.d_e) ? (unsigned)(k - g.d_v) * 4u : kOOB;
#pragma unroll
        for (int j = 0; j < JR * RT_E; ++j)
            a_grp[j] = __builtin_amdgcn_raw_buffer_load_b64(rVg, gemm::join_off(ro1[j], k1o), 0, 0) |
                       __builtin_amdgcn_raw_buffer_load_b64(rEg, gemm::join_off(ro2[j], k2o), 0, 0);
    }
    unsigned pf_word[4] = {0u, 0u, 0u, 0u};   // (the L2 warm-up below, behind the K1 contraction: loaded there, looked at only when the kernel ends)
#if defined(DMPNN_META_STAMPS)
    stamp();  // m1: tile table + index loads issued, operand gather in flight
#endif
    if (tid < BM) { revl[tid] = revl_v; asrc[tid] = asrc_v; if (lean) aor[tid] = aor_v; }
    if (tid <= BA) rp[tid] = rp_v;
    if (tid < 8) maxbits[tid] = 0u;
    __syncthreads();
#if defined(DMPNN_META_STAMPS)
    stamp();  // m2: index values arrived, LDS metadata written
#endif
    if (row_bad) atomicOr(&maxbits[5], 1u);
    if (!lean && tid < na)
        for (int r = rp[tid]; r < rp[tid + 1]; ++r) aor[r] = tid;
    __syncthreads();
#if defined(DMPNN_META_STAMPS)
    stamp();  // m3: atom-of-row table complete
#endif
    // Incidence fragments (B operands of the segment MFMAs, constant over the depth loop).  The k index of
    // those MFMAs runs over the tile's rows in the order the C/D fragments of a contraction already hold
    // them: k-step 0, lane group lg, slot s -> row lg*4+s (s<4) | 16+lg*4+(s-4);  k-step 1 -> row 32+lg*4+s
    // (s<4) | none.  Fragments 0..5: message, C[r'][r] = 1 iff r enters the source atom of r' and r != rev(r')
    // (message_passing/base.py:144-146 with the reverse edge's cancelling term left out of the sum);
    // 6..9: aggregate, C[a][r] = 1 iff r enters atom a (base.py:208-211).
    // A lane's eight k rows depend on (k-step, lg) only: their atoms are read once (three 16-byte LDS reads), a fragment is then
    // one column lookup + eight integer compares, packed as f16 bit patterns (1.0 = 0x3C00, -1.0 = 0xBC00).
    int ar[12];
    {
        const int4 q0 = *reinterpret_cast<const int4*>(aor + lg * 4), q1 = *reinterpret_cast<const int4*>(aor + 16 + lg * 4),
                   q2 = *reinterpret_cast<const int4*>(aor + 32 + lg * 4);
        const int qa[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int i = 0; i < 12; ++i) ar[i] = ((i >> 2) * 16 + lg * 4 + (i & 3)) < nrows ? qa[i] : -2;  // (a row past the tile matches no atom)
    }
    for (int f = wave; f < 10; f += NW) {
        const bool agg = f >= 6;
        const int ff = agg ? f - 6 : f, jt = ff >> 1, ks = ff & 1;
        const int j = jt * 16 + li;
        int a_t = -1, rv = -1;
        if (agg) {
            a_t = j < na ? j : -1;
        } else if (j < nrows) {
            rv = revl[j];
            a_t = lean ? asrc[j] : aor[rv];  // source atom of r' (CSR plan: the graph is symmetric, src r' = dst rev r')
        }
        if (G.atom_de) rv = -1;  // atom messages, mixins.py:25-30: the plain sum — no reverse-edge term
        unsigned hb[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int row = ks == 0 ? (sl < 4 ? lg * 4 + sl : 16 + lg * 4 + (sl - 4)) : (sl < 4 ? 32 + lg * 4 + sl : -2);
            const int av = ks == 0 ? ar[sl] : (sl < 4 ? ar[8 + sl] : -2);
            // [r enters src r'] - [r = rev r']: 0/1 for a symmetric graph (the reverse edge enters src r'), -1/0/1 otherwise
            const bool in = av == a_t, isrev = row == rv;
            hb[sl] = in ? (isrev ? 0u : 0x3C00u) : (isrev ? 0xBC00u : 0u);
        }
        const u32x4 pk = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
        cfrag[f * 64 + lane] = __builtin_bit_cast(h8, pk);
    }
    __syncthreads();

    // maximum of a non-negative float over the wave (their bit patterns order like the values): DPP within
    // the rows of 16 lanes, then the four row results through scalar registers.  Uniform result.
    auto wave_max = [&](float v) -> float {
        int u = (int)__float_as_uint(v);
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x141, 0xf, 0xf, true));  // row_half_mirror
        u = max(u, __builtin_amdgcn_update_dpp(0, u, 0x140, 0xf, 0xf, true));  // row_mirror
        const int m = max(max(__builtin_amdgcn_readlane(u, 0), __builtin_amdgcn_readlane(u, 16)),
                          max(__builtin_amdgcn_readlane(u, 32), __builtin_amdgcn_readlane(u, 48)));
        return __uint_as_float((unsigned)m);
    };
    // tile maximum of |x| over per-thread values -> exact power-of-two scale.  The LDS word rotates over
    // four slots: call p uses slot p & 3 and re-arms slot (p + 2) & 3 behind its barrier (last read before
    // barrier p - 1, next written after barrier p + 1), so no extra barrier is spent on the reset.
    int scale_phase = 0;
    float tile_mx = 0.f;   // the maximum the last tile_scale call saw (0: an all-zero tile — its kept split rows say so in their tails)
    auto tile_scale = [&](float local_max) -> float {
        const int slot = scale_phase & 3;
        local_max = wave_max(local_max);
        if (lane == 0) atomicMax(&maxbits[slot], __float_as_uint(local_max));  // non-negative floats order like their bits
        __syncthreads();
        const float mx = __uint_as_float(maxbits[slot]);
        tile_mx = mx;
        if (tid == 0) maxbits[(slot + 2) & 3] = 0u;
        ++scale_phase;
        return scale_for(mx);
    };

    // ---- the tile's body, per wave class: WL = column tiles of this wave, ct0() = its first one (NW = 4: WL = WN for every wave) ----
    auto body = [&](auto wl_c) __attribute__((always_inline)) {
    constexpr int WL = decltype(wl_c)::value;
    auto ct0 = [&]() -> int { return (NW == 8 && WL == 2) ? 2 * wave + 4 : WL * wave; };
    // ---- weight fragments: straight from L2 to registers --------------------------------------------
    // The four waves of a workgroup own disjoint column ranges, so a weight element is used by exactly one
    // wave: staging the weight tile in LDS would buy no reuse and cost a write + a read + a barrier per
    // chunk.  Lane (li, lg) of column tile ct reads its own fragment of chunk c from the pre-split layout:
    // hi 16 B at [col][c][lg*16], lo at +64 (col = (ct0 + ct)*16 + li; col >= N is out of range: 0).
    auto load_bfrags = [&](rsrc_t rW, const unsigned (&offB)[WL], int c, h8 (&bh)[WL], h8 (&bl)[WL]) {
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct] + (unsigned)c * 2048u, 0, 0);
            bh[ct] = __builtin_bit_cast(h8, vh);
            if constexpr (!LP) {
                const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rW, offB[ct] + (unsigned)c * 2048u + 1024u, 0, 0);
                bl[ct] = __builtin_bit_cast(h8, vl);
            }
        }
    };
    auto bfrag_offsets = [&](const SplitW& W, unsigned (&offB)[WL]) {
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct)  // fragment-major layout [column tile][chunk][hi|lo][lane][16 B]
            offB[ct] = (unsigned)(ct0() + ct) * (unsigned)(W.nc * 2048) + (unsigned)lane * 16u;
    };
    // Round 6: a kept message leaves as WHOLE ROWS of the split A tile — the tile in LDS is the kept row format (chunk pairs + the
    // 16-byte tail with the tile's scale), so once it is complete (the contraction's barrier) every thread copies 16-byte pieces, 1 KiB
    // per wave instruction, instead of each lane storing its own 8-byte pieces behind the segment MFMAs (18 scattered store instructions
    // per wave and step: "split written" 1.3 k -> 4.7-5.0 k cycles of the training forward's phase stamps)
    unsigned char* rows_pending = nullptr;
    int rows_pending_n = 0;
    auto flush_rows = [&]() {
        if constexpr (KEEP) {
            if (rows_pending) {   // (uniform)
                constexpr int PPR = TS / 16;   // 16-byte pieces of a row, tail included
                for (int it = tid; it < rows_pending_n * PPR; it += KT) {
                    const int r = it / PPR, pc = it - r * PPR;
                    *reinterpret_cast<u32x4*>(rows_pending + (long long)r * TS + pc * 16) = *reinterpret_cast<const u32x4*>(T16 + r * TS + pc * 16);
                }
                rows_pending = nullptr;
            }
        }
    };
    // ---- one contraction: acc[RT][WL] += (A s_A) . (W s_W)^T in the split domain -------------------
    // A fragments straight from a split tile in LDS (static during the contraction: NO barrier in the main
    // loop): row stride `astride` bytes, chunk c at +c*128 as [hi 32 halfs | lo 32 halfs].  Weight chunks
    // wc0 .. wc0 + n_chunks - 1 of W.
    auto contract = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL], const unsigned char* Ab, int astride, int n_chunks,
                        int wc0, const SplitW& W) {
        constexpr int RT = decltype(rt_c)::value;
        const rsrc_t rW = gemm::make_rsrc(W.p, (unsigned)(((N + 15) / 16) * W.nc * 2048));  // column tiles beyond N: out of range, 0
        unsigned offB[WL];
        bfrag_offsets(W, offB);
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) offB[ct] += (unsigned)wc0 * 2048u;
        auto read_afrags = [&](int c, h8 (&ah)[RT], h8 (&al)[RT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned char* p = Ab + (rt * 16 + li) * astride + c * 128 + lg * 16;
                ah[rt] = *reinterpret_cast<const h8*>(p);
                if constexpr (!LP) al[rt] = *reinterpret_cast<const h8*>(p + 64);
            }
        };
        // ONE set of weight fragments, used as a ring over the column tiles (the register budget of two workgroups per CU is
        // 256): the three products of column tile ct are issued together and its fragments of chunk c + 1 are requested right
        // behind them (distance: the 9 RT MFMAs of each of the other column tiles, exact vmcnt in the compact loop); the A
        // fragments of chunk c + 1 are read from LDS under the last-but-one column tile.  What latency is left uncovered is the
        // other workgroup's to fill.
        h8 a0h[RT], a0l[RT], a1h[RT], a1l[RT];
        if constexpr (NW == 8 && DMPNN_TILE8_BRING == 2) {
            // 8 waves: a wave owns 3 or 2 column tiles, so the one-set ring's prefetch distance (the other tiles' 9 RT MFMAs each) is
            // 18 / 9 MFMAs — under the L2 latency.  The registers the narrower accumulators free hold a SECOND set: chunk c + 2 is
            // requested behind the products of chunk c (distance: a whole chunk more)
            h8 b0h[WL], b0l[WL], b1h[WL], b1l[WL];
            load_bfrags(rW, offB, 0, b0h, b0l);
            {   // (a select, not a branch: two paths into the loop with different load queues would make hipcc wait for the shorter one)
                const bool two = n_chunks > 1;
#pragma unroll
                for (int ct = 0; ct < WL; ++ct) {
                    b1h[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, two ? offB[ct] + 2048u : kOOB, 0, 0));
                    if constexpr (!LP) b1l[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, two ? offB[ct] + 3072u : kOOB, 0, 0));
                }
            }
            __syncthreads();  // the split A tile is complete
            flush_rows();
            launder();
            read_afrags(0, a0h, a0l);
            auto chunk2 = [&](int c, h8 (&ah)[RT], h8 (&al)[RT], h8 (&nah)[RT], h8 (&nal)[RT], h8 (&bh)[WL], h8 (&bl)[WL]) {
                const bool more2 = c + 2 < n_chunks;
#pragma unroll
                for (int ct = 0; ct < WL; ++ct) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bh[ct], acc[rt][ct], 0, 0, 0);
                    if constexpr (!LP) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bh[ct], acc[rt][ct], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned o = more2 ? offB[ct] + (unsigned)(c + 2) * 2048u : kOOB;  // (past the last chunk: out of range, 0)
                    bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
                    if constexpr (!LP) bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, more2 ? o + 1024u : kOOB, 0, 0));
                    if (ct == (WL > 1 ? WL - 2 : 0)) read_afrags(c + 1 < n_chunks ? c + 1 : c, nah, nal);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // whole pairs in the loop, an odd last chunk behind it: a skip INSIDE the loop would make hipcc merge the two paths' load
            // queues at the back edge and wait for the youngest set every chunk (vmcnt(3) where 7 are legitimately in flight)
            int c = 0;
#pragma nounroll
            for (; c + 1 < n_chunks; c += 2) {
                chunk2(c, a0h, a0l, a1h, a1l, b0h, b0l);
                chunk2(c + 1, a1h, a1l, a0h, a0l, b1h, b1l);
            }
            if (c < n_chunks) chunk2(c, a0h, a0l, a1h, a1l, b0h, b0l);
            return;
        }
        h8 bh[WL], bl[WL];
        load_bfrags(rW, offB, 0, bh, bl);
        __syncthreads();  // the split A tile is complete
        flush_rows();
        launder();
        read_afrags(0, a0h, a0l);
        auto chunk = [&](int c, h8 (&ah)[RT], h8 (&al)[RT], h8 (&nah)[RT], h8 (&nal)[RT]) {
            const bool more = c + 1 < n_chunks;
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bh[ct], acc[rt][ct], 0, 0, 0);
                if constexpr (!LP) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bh[ct], acc[rt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned o = more ? offB[ct] + (unsigned)(c + 1) * 2048u : kOOB;  // (past the last chunk: out of range, 0)
                bh[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
                if constexpr (!LP) bl[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, more ? o + 1024u : kOOB, 0, 0));
                if (ct == (WL > 1 ? WL - 2 : 0)) read_afrags(more ? c + 1 : c, nah, nal);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma nounroll
        for (int c = 0; c < n_chunks; c += 2) {
            chunk(c, a0h, a0l, a1h, a1l);
            if (c + 1 < n_chunks) chunk(c + 1, a1h, a1l, a0h, a0l);
        }
        // (no trailing barrier: every writer of the A tile sits behind the barrier of a tile_scale call)
    };

    // ---- A operand from global memory, in groups of up to 4 k-chunks (128 columns) ---------------------
    // Item j of a thread: row wave + NW j, column pair `lane` of the group (k = 128 grp + 2 lane), so one
    // wave instruction reads 512 contiguous bytes of one (gathered) row.  The group is held in registers:
    // its maximum gives the tile scale, then it is split into the LDS tile Ag (overlays T16 / T) — the
    // operand is read from memory exactly once and the contraction over it is the barrier-free one above.
    constexpr int TSG = 4 * 128 + 16;
    unsigned char* Ag = lds;
    auto ga_load = [&](auto rt_c, auto has_a2_c, int grp, int K1, int K2, rsrc_t rA1, rsrc_t rA2,
                       const unsigned (&ro1)[JR * decltype(rt_c)::value], const unsigned (&ro2)[JR * decltype(rt_c)::value],
                       u32x2 (&v)[JR * decltype(rt_c)::value]) {
        constexpr int J = JR * decltype(rt_c)::value;
        constexpr bool HAS_A2 = decltype(has_a2_c)::value;
        const int k = grp * 128 + lane * 2;
        const unsigned k1o = k < K1 ? (unsigned)k * 4u : kOOB;
        const unsigned k2o = (k >= K1 && k < K1 + K2) ? (unsigned)(k - K1) * 4u : kOOB;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            v[j] = __builtin_amdgcn_raw_buffer_load_b64(rA1, gemm::join_off(ro1[j], k1o), 0, 0);
            if constexpr (HAS_A2) v[j] = v[j] | __builtin_amdgcn_raw_buffer_load_b64(rA2, gemm::join_off(ro2[j], k2o), 0, 0);
        }
    };
    // scale of the group, rescale of what `acc` holds from the previous scale, split + store
    auto ga_stage = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL], const u32x2 (&v)[JR * decltype(rt_c)::value], float s_prev) -> float {
        constexpr int RT = decltype(rt_c)::value, J = JR * RT;
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) mx = fmaxf(mx, fmaxf(fabsf(__uint_as_float(v[j].x)), fabsf(__uint_as_float(v[j].y))));
        const float s = tile_scale(mx);  // (barrier: every wave is past its reads of the LDS tiles)
        if (s_prev != 0.f && s_prev != s) {  // bring the accumulated part into this group's scale (exact: powers of two)
            const float f = s / s_prev;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rt][ct][r] *= f;
        }
        launder();
#pragma unroll
        for (int j = 0; j < J; ++j) {
            unsigned hi, lo;
            split2(__uint_as_float(v[j].x), __uint_as_float(v[j].y), s, hi, lo);
            unsigned char* p = Ag + (wave + NW * j) * TSG + (lane >> 4) * 128 + (lane & 15) * 4;
            *reinterpret_cast<unsigned*>(p) = hi;
            if constexpr (!LP) *reinterpret_cast<unsigned*>(p + 64) = lo;
        }
        return s;
    };
    auto zero_acc = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL]) {
        constexpr int RT = decltype(rt_c)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // per-lane column constants of a contraction (inverse weight scale, bias), fetched BEFORE the
    // contraction so their latency hides under it
    struct ColConst { float isw[WL], bv[WL]; };
    auto col_consts = [&](const float* inv_sW, const float* bias) -> ColConst {
        ColConst cc;
        launder();
        const float* bp = bias ? bias : inv_sW;
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col = (ct0() + ct) * 16 + li;
            const bool okc = col < N;
            cc.isw[ct] = inv_sW[okc ? col : 0];
            const float braw = bp[okc ? col : 0];
            cc.bv[ct] = (okc && bias) ? braw : 0.f;
        }
        return cc;
    };
    // split domain -> fp32:  z = acc / (sA sW[col]) + bias[col]
    auto unscale = [&](auto rt_c, f32x4 (&acc)[decltype(rt_c)::value][WL], float inv_sA, const ColConst& cc) {
        constexpr int RT = decltype(rt_c)::value;
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const float isw = cc.isw[ct] * inv_sA;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][ct][r] = acc[rt][ct][r] * isw + cc.bv[ct];
        }
    };
    // y = tau(z [+ res]) elementwise on C/D fragments
    auto act_frags = [&](auto rt_c, auto use_res_c, f32x4 (&z)[decltype(rt_c)::value][WL], const f32x4 (&res)[decltype(rt_c)::value][WL]) {
        constexpr int RT = decltype(rt_c)::value;
        constexpr bool USE_RES = decltype(use_res_c)::value;
        if constexpr (simple_act) {
#if !defined(DMPNN_RELU_SELECT)
            if (neg_slope == 0.f) {   // (uniform) ReLU: ONE v_maximum3_f32(v, 0, 0) per element instead of multiply + compare + select + add.  IEEE-754
                                      // maximum: a NaN stays NaN (loud), -0 -> +0, and -inf -> 0 as torch.relu has it (the select form made it NaN)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = USE_RES ? res[rt][ct][r] + z[rt][ct][r] : z[rt][ct][r];
                            z[rt][ct][r] = __builtin_elementwise_maximum(v, 0.f);
                        }
                return;
            }
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = USE_RES ? res[rt][ct][r] + z[rt][ct][r] : z[rt][ct][r];
                        z[rt][ct][r] = (v > 0.f ? v : neg_slope * v) + 0.f;
                    }
        } else {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = USE_RES ? res[rt][ct][r] + z[rt][ct][r] : z[rt][ct][r];
                        z[rt][ct][r] = apply_act(v, g.act, slope);
                    }
        }
    };
    // active dropout on C/D fragments (training forward only): element (row0 + rt 16 + lg 4 + r, column) is kept iff its hash
    // clears the threshold, and scaled by 1 / (1 - p)   (base.py:139,182)
    // (edge sites are keyed on the CALLER's edge id — the same mask whatever plan the step runs on: under a CSR plan the tile's row r
    //  is edge perm[rs + r], under a tile plan edge rs + r)
    auto dropout_frags = [&](auto rt_c, f32x4 (&y)[decltype(rt_c)::value][WL], unsigned site, int row0, bool edge_rows) {
        constexpr int RT = decltype(rt_c)::value;
        if constexpr (KEEP) {
            if (g.drop_thr) {  // (uniform)
                launder();
                unsigned rowid[RT][4];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = rt * 16 + lg * 4 + r;
                        rowid[rt][r] = (edge_rows && !lean) ? (unsigned)g.perm[rs + (lr < nrows ? lr : 0)] : (unsigned)(row0 + lr);
                    }
#pragma unroll
                for (int ct = 0; ct < WL; ++ct) {
                    const unsigned col = (unsigned)((ct0() + ct) * 16 + li);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned row = rowid[rt][r];
                            const bool keep = drop_hash(g.seed_lo, g.seed_hi, site, row, col) >= g.drop_thr;
                            y[rt][ct][r] = keep ? y[rt][ct][r] * g.drop_scale : 0.f;
                        }
                }
            }
        }
    };
    auto frag_to_tile = [&](auto rt_c, const f32x4 (&y)[decltype(rt_c)::value][WL]) {
        constexpr int RT = decltype(rt_c)::value;
        launder();
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col = (ct0() + ct) * 16 + li;
            if (col < N) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(rt * 16 + lg * 4 + r) * LDC + col] = y[rt][ct][r];
            }
        }
    };
    // sign bits of C/D fragments -> keep_bits (see Mega16K): 4 RT WN ballots, gathered into the lanes by v_writelane, one 8-byte
    // store per lane — no LDS, no barrier, 2 KB per tile instead of the tile's fp32 rows
    auto store_bits = [&](auto rt_c, const f32x4 (&y)[decltype(rt_c)::value][WL], int slot) {
        constexpr int RT = decltype(rt_c)::value;
        static_assert(RT * WL * 4 <= 64, "one word per lane");
        unsigned lo = 0u, hi = 0u;
        static_for<0, RT * WL * 4>([&](auto ic) {   // (the lane select of v_writelane_b32 must be an inline constant: one SGPR operand per VALU instruction)
            constexpr int idx = decltype(ic)::value, r = idx & 3, ct = (idx >> 2) % WL, rt = (idx >> 2) / WL;
            const unsigned long long b = __ballot(y[rt][ct][r] > 0.f);
            unsigned l = lo, h = hi;   // (asm operands must be locals of this lambda)
            const unsigned bl = (unsigned)b, bh = (unsigned)(b >> 32);
            // (s_nop: an SGPR written by the v_cmp right in front is not yet readable by v_writelane — measured: the low halves came
            //  out stale without it, scripts/dbg_bits.py; the compiler's hazard recognizer does not look into inline assembly)
            asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(l) : "s"(bl), "n"(idx));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(h) : "s"(bh), "n"(idx));
            lo = l; hi = h;
        });
        launder();
        if (lane < RT * WL * 4) {
            // the word of (row tile rt, column tile gct = ct0 + ct, register r) sits where the 4-wave form puts it — wave gct / WN, slot
            // (rt WN + gct % WN) 4 + r — whatever the wave split: the backward tile kernels read ONE layout
            const int q = lane >> 2, gct = ct0() + q % WL, rt_ = q / WL;
            const int word = NW == 4 ? wave * 64 + lane : (gct / WN) * 64 + (rt_ * WN + gct % WN) * 4 + (lane & 3);
            G.keep_bits[(long long)slot * G.bits_slot + (long long)t * 256 + word] = ((unsigned long long)hi << 32) | lo;
        }
    };
    auto tile_to_global = [&](float* dst, long long row0, int ld, int n_r) {
        launder();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + KT * j;
            const int r = it / QN, q = it - r * QN;
            if (r < n_r && q < qn)
                store_keep4(dst + (row0 + r) * ld + 4 * q, *reinterpret_cast<const float4*>(T + r * LDC + 4 * q));
        }
    };
    // message / aggregate as MFMAs on the contraction's own C/D fragments (no LDS round trip, no barrier):
    //   last == false:  M[r']  = sum_r C[r'][r] H[r]   (= S[src r'] - H[rev r'], mixins.py:11-18)
    //   last == true :  Mv[a]  = sum_r C[a][r]  H[r]   (base.py:208-211)
    // computed transposed, D[col][r'] = sum_k H^T[col][k] C^T[k][r']: lane (li, lg) of a C/D fragment holds
    // column li of rows lg*4.. of each row tile, which is exactly the A-operand slice (row li, k = lg*8..) of
    // H^T once the k order is the one of the incidence fragments above.  H is split per wave
    // (x s_H = hi + lo, 22 bits relative to the wave maximum), C is exact in f16, accumulation fp32.
    // The result lands as 4 consecutive columns of row r' (li) per lane: split with the tile scale of the
    // next contraction and written to T16 as 8-byte pieces; the fp32 copy streams to `keep`.
    static_assert(RT_E == 3 && RT_A == 2, "segment MFMAs are laid out for 48-row / 32-atom tiles");
    auto segment_mfma = [&](const f32x4 (&H)[RT_E][WL], bool last, float* keep, int keep_ld, unsigned char* keep_rows) -> float {
        launder();
        float hm = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) hm = fmaxf(hm, fabsf(H[rt][ct][r]));
        const float sH = scale_for(wave_max(hm));
        const h8* Cf = cfrag + lane + (last ? 6 * 64 : 0);
        h8 cf[RT_E][2];
#pragma unroll
        for (int jt = 0; jt < RT_E; ++jt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) cf[jt][ks] = Cf[((jt < RT_A || !last ? jt : 0) * 2 + ks) * 64];
        f32x4 m[WL][RT_E];
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            h8 ah0, al0, ah1, al1;
            {   // k-step 0: rows lg 4 .. of row tiles 0 | 1, k-step 1: of row tile 2 | none
                unsigned wh0[4], wl0[4], wh1[4] = {0u, 0u, 0u, 0u}, wl1[4] = {0u, 0u, 0u, 0u};
                split2(H[0][ct][0], H[0][ct][1], sH, wh0[0], wl0[0]);
                split2(H[0][ct][2], H[0][ct][3], sH, wh0[1], wl0[1]);
                split2(H[1][ct][0], H[1][ct][1], sH, wh0[2], wl0[2]);
                split2(H[1][ct][2], H[1][ct][3], sH, wh0[3], wl0[3]);
                split2(H[2][ct][0], H[2][ct][1], sH, wh1[0], wl1[0]);
                split2(H[2][ct][2], H[2][ct][3], sH, wh1[1], wl1[1]);
                ah0 = __builtin_bit_cast(h8, u32x4{wh0[0], wh0[1], wh0[2], wh0[3]}); al0 = __builtin_bit_cast(h8, u32x4{wl0[0], wl0[1], wl0[2], wl0[3]});
                ah1 = __builtin_bit_cast(h8, u32x4{wh1[0], wh1[1], wh1[2], wh1[3]}); al1 = __builtin_bit_cast(h8, u32x4{wl1[0], wl1[1], wl1[2], wl1[3]});
            }
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                if (jt < RT_A || !last) {
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, cf[jt][0], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, cf[jt][1], z, 0, 0, 0);
                    if constexpr (!LP) {
                        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, cf[jt][0], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, cf[jt][1], z, 0, 0, 0);
                    }
                }
                m[ct][jt] = z;
            }
        }
        stamp();  // s1: segment MFMAs
        const float isH = rcp_pow2_exact(sH);
        float mx = 0.f;
#pragma unroll
        for (int ct = 0; ct < WL; ++ct)
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[ct][jt][r] *= isH;
                    mx = fmaxf(mx, fabsf(m[ct][jt][r]));
                }
        const float s = tile_scale(mx);  // (contains a barrier: every wave is past its reads of T16)
        stamp();  // s2: tile scale
        const int n_keep = last ? na : nrows;
        const long long keep0 = last ? va : rs;
#pragma unroll
        for (int ct = 0; ct < WL; ++ct) {
            const int col4 = (ct0() + ct) * 16 + lg * 4;
#pragma unroll
            for (int jt = 0; jt < RT_E; ++jt) {
                if (jt < RT_A || !last) {
                    const int row = jt * 16 + li;
                    const float4 v = make_float4(m[ct][jt][0], m[ct][jt][1], m[ct][jt][2], m[ct][jt][3]);
                    h4 hi, lo;
                    split4(v, s, hi, lo);
                    unsigned char* p = T16 + row * TS + (col4 >> 5) * 128 + (col4 & 31) * 2;
                    *reinterpret_cast<h4*>(p) = hi;
                    if constexpr (!LP) *reinterpret_cast<h4*>(p + 64) = lo;
                    if constexpr (KEEP) {
                        // (keep_rows: the kept split rows are copied out of this tile as whole rows — flush_rows)
                        if (!keep_rows && keep && row < n_keep && col4 < N) store_keep4(keep + (keep0 + row) * keep_ld + col4, v);
                    }
                }
            }
        }
        if constexpr (KEEP) {
            if (keep_rows) {   // (uniform)
                if (wave == 0 && lg == 0) {   // the rows' tails (the tile's scale, the all-zero flag) into the tile's own tail slots
#pragma unroll
                    for (int jt = 0; jt < RT_E; ++jt)
                        *reinterpret_cast<float4*>(T16 + (jt * 16 + li) * TS + (TS - 16)) = make_float4(s, tile_mx > 0.f ? 0.f : 1.f, 0.f, 0.f);
                }
                rows_pending = keep_rows + keep0 * TS;   // (G.tsr == TS: launch_mega16_forward checks)
                rows_pending_n = n_keep;
            }
        }
        return s;
    };

    using RE = std::integral_constant<int, RT_E>;
    using RA = std::integral_constant<int, RT_A>;
    const int T_steps = g.depth;

    // ================= K1: H0 = W_i [V[src] || E] =================
    f32x4 h0[RT_E][WL];
    {
        stamp();  // 1: metadata done
        const ColConst cc = col_consts(G.Wi.inv_scale, g.b_i);
        zero_acc(RE{}, h0);
        float s_prev = 0.f;
        for (int grp = 0; grp * 4 < G.Wi.nc; ++grp) {
            if (grp > 0) ga_load(RE{}, T_{}, grp, g.d_v, g.d_e, rVg, rEg, ro1, ro2, a_grp);
            s_prev = ga_stage(RE{}, h0, a_grp, s_prev);
            if (grp == 0) stamp();  // 2: init A staged
            const int ncg = G.Wi.nc - grp * 4 < 4 ? G.Wi.nc - grp * 4 : 4;
            contract(RE{}, h0, Ag, TSG, ncg, grp * 4, G.Wi);
        }
        unscale(RE{}, h0, rcp_pow2_exact(s_prev), cc);
        stamp();  // 3: K1 contraction
    }
    // Round 6: warm THIS XCD's L2 with W_h and W_o's message part behind the K1 contraction (its own weight requests are all back), under the
    // first message's VALU / LDS phases.  Every tile of a launch reaches its first update contraction at about the same time and all of them
    // miss on the same lines (a forward splits the weights afresh: they are in no L2): the stamps of the hi-halves form — whose contractions
    // are short enough to show it — had the FIRST use of W_h at 7.7 k cycles and the second at 4.6 k, the first use of W_o's message part
    // at 7.5 k.  Workgroups go round the XCDs (t mod 8), so the t / 8-th workgroup of an XCD touches the t / 8-th share of the lines: one
    // 4-byte load per thread and 128-byte line, consumed by nothing (the word below is never read).
    // (Also measured, profiles/r06_l2_warm_variants.txt: the same loads as inline assembly into one sink register — invisible to hipcc's wait
    //  counts, the first vmcnt(0) behind them waited for the whole burst, +2 us per launch; as plain global loads — hipcc cannot count a FLAT
    //  load in order and every later wait became vmcnt(0), the update contraction 8.5 k -> 13.7 k cycles; W_i touched at kernel entry and
    //  W_o's atom part with the two above — +0.3 us on every launch, the gain at 512 molecules down from 0.9 to 0.25 us.  Hence buffer loads,
    //  these two matrices, behind K1.)
#if !defined(DMPNN_NO_L2_WARM)
    // (measured, same box, module forward with | without: 512 molecules / 230 tiles 39.2-39.4 | 39.9 us and 30.4 | 32.9 us on the hi halves;
    //  64 molecules / 29 tiles 38.1 | 37.4 and 1 024 molecules / 460 tiles 65.7 | 64.8: a launch of few tiles or of two rounds loses — hence the window)
#if !defined(DMPNN_L2_WARM_MIN_TILES)
#define DMPNN_L2_WARM_MIN_TILES 96
#define DMPNN_L2_WARM_MAX_TILES 256
#endif
    if (g.depth > 1 && hdr_tiles >= DMPNN_L2_WARM_MIN_TILES && hdr_tiles <= DMPNN_L2_WARM_MAX_TILES && t < 256) {   // (uniform)
        launder();
        const unsigned nl = (unsigned)(((N + 15) / 16) * G.Wh.nc * 16);   // 128-byte lines of one split matrix (2 KiB per column tile and chunk)
        const rsrc_t rWh = gemm::make_rsrc(G.Wh.p, nl * 128u), rWo = gemm::make_rsrc(G.WoM.p, nl * 128u);
        const unsigned share = (unsigned)(((hdr_tiles < 256 ? hdr_tiles : 256) + 7) >> 3) * (unsigned)KT;
        unsigned line = (unsigned)(t >> 3) * (unsigned)KT + (unsigned)tid;
#pragma unroll
        for (int k = 0; k < 2; ++k, line += share) {
            const unsigned oh = line < nl ? line * 128u : kOOB, oo = (line >= nl && line < 2u * nl) ? (line - nl) * 128u : kOOB;
            pf_word[2 * k] = __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(rWh, oh, 0, 0));
            pf_word[2 * k + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(rWo, oo, 0, 0));
        }
    }
#endif
    if (KEEP && G.keep_bits) {  // training: the backward pass needs tau'(H0) — for a ReLU-class tau the sign, as bits
        store_bits(RE{}, h0, 0);
    } else if (KEEP && g.H0) {  // ... else the pre-activation itself
        __syncthreads();  // (the fp32 tile may overlay the K1 operand tile)
        frag_to_tile(RE{}, h0);
        __syncthreads();
        tile_to_global(g.H0, rs, g.ldh, nrows);
    }
    float sA;
    {
        f32x4 y[RT_E][WL];
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int ct = 0; ct < WL; ++ct) y[rt][ct] = h0[rt][ct];
        act_frags(RE{}, F_{}, y, y);
        sA = segment_mfma(y, T_steps == 1, T_steps == 1 ? g.Mv : (G.Mrows ? nullptr : g.Ms), g.ldh, T_steps == 1 ? nullptr : G.Mrows);
    }
    stamp();  // 4: K1 epilogue + first message
    if (G.atom_de) {
        // (uniform) atom messages, mixins.py:25-30: M[r'] = sum_{r enters src r'} [H[r] || E[r]].  The bond-feature half does not change
        // over the depth loop: x[r'] = W_h[:, N:] (sum E)[src r'] is formed ONCE per tile and joins the residual of every update
        // (the first activation above saw the pure H0 = W_i V[src], base.py:200).
        const int de = G.atom_de;
        float* Et = reinterpret_cast<float*>(cfrag + 10 * 64);   // [BM][kAtomK] E rows of the tile
        float* SE = Et + BM * kAtomK;                            // [BA][kAtomK] per-atom sums of the incoming rows' E
        for (int i = tid; i < BM * kAtomK; i += KT) {
            const int r = i / kAtomK, k = i - r * kAtomK;
            float v = 0.f;
            if (r < nrows && k < de) {
                const long long e = lean ? (long long)(rs + r) : (long long)g.perm[rs + r];
                v = g.E[e * g.lde + k];
            }
            Et[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < BA * kAtomK; i += KT) {      // increasing row order = the reference's sequential scatter order
            const int a_ = i / kAtomK, k = i - a_ * kAtomK;
            float sum = 0.f;
            if (a_ < na && k < de)
                for (int r = 0; r < nrows; ++r)
                    if (aor[r] == a_) sum += Et[r * kAtomK + k];
            SE[i] = sum;
        }
        __syncthreads();
        launder();
        if constexpr (KEEP) {
            if (G.atom_me) {  // (uniform) what the weight-gradient product of W_h[:, N:] contracts gZ^(t) with — the same rows for every step
                for (int i = tid; i < nrows * kAtomK; i += KT) {
                    const int r = i / kAtomK, k = i - r * kAtomK;
                    const float v = SE[(lean ? asrc[r] : aor[revl[r]]) * kAtomK + k];
                    for (int sl = 0; sl < T_steps - 1; ++sl) G.atom_me[(long long)sl * G.me_slot + (long long)(rs + r) * kAtomK + k] = v;
                }
            }
        }
        // A fragments of ME[r'] = SE[src r'] (row rt 16 + li, k = lg 8 .. lg 8 + 7 of the one chunk) straight into registers
        float xv[RT_E][8];
        float mx = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt) {
            const int row = rt * 16 + li;
            const int a_s = row < nrows ? (lean ? asrc[row] : aor[revl[row]]) : -1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = lg * 8 + j;
                xv[rt][j] = (a_s >= 0 && k < kAtomK) ? SE[a_s * kAtomK + (k < kAtomK ? k : 0)] : 0.f;
                mx = fmaxf(mx, fabsf(xv[rt][j]));
            }
        }
        const float sE = tile_scale(mx);
        h8 eh[RT_E], el[RT_E];
#pragma unroll
        for (int rt = 0; rt < RT_E; ++rt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = xv[rt][j] * sE;
                const _Float16 hi = (_Float16)x;
                eh[rt][j] = hi; el[rt][j] = (_Float16)(x - (float)hi);
            }
        unsigned offE[WL];
        bfrag_offsets(G.WhE, offE);
        h8 wh[WL], wl[WL];
        load_bfrags(gemm::make_rsrc(G.WhE.p, (unsigned)(((N + 15) / 16) * G.WhE.nc * 2048)), offE, 0, wh, wl);
        const ColConst ce = col_consts(G.WhE.inv_scale, nullptr);
#pragma unroll
        for (int ct = 0; ct < WL; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT_E; ++rt) {
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[rt], wh[ct], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[rt], wl[ct], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(el[rt], wh[ct], z, 0, 0, 0);
                const float isw = ce.isw[ct] / sE;
#pragma unroll
                for (int r = 0; r < 4; ++r) h0[rt][ct][r] += z[r] * isw;
            }
    }

    // ================= K3 x (depth - 1): H = tau(H0 + W_h M) =================
    for (int step = 1; step < T_steps; ++step) {
        f32x4 acc[RT_E][WL];
        zero_acc(RE{}, acc);
        const ColConst cc = col_consts(G.Wh.inv_scale, g.b_h);
        contract(RE{}, acc, T16, TS, (N + BK - 1) / BK, 0, G.Wh);
        stamp();  // 5, 7, ...: update contraction
        unscale(RE{}, acc, rcp_pow2_exact(sA), cc);
        act_frags(RE{}, T_{}, acc, h0);  // tau(H0 + W_h(M)): base.py:141
        dropout_frags(RE{}, acc, (unsigned)(step - 1), rs, true);  // dropout(H_t): base.py:139 (training, p > 0)
        stamp();  // E: unscale + tau
        if (KEEP && G.keep_bits) {
            store_bits(RE{}, acc, step);
        } else if (KEEP && g.Hs) {
            __syncthreads();  // (the fp32 tile overlays the split A tile: every wave is past its contraction)
            frag_to_tile(RE{}, acc);
            __syncthreads();
            tile_to_global(g.Hs + (long long)(step - 1) * g.slot, rs, g.ldh, nrows);
        }
        const bool last = step == T_steps - 1;
        sA = segment_mfma(acc, last, last ? g.Mv : ((g.Ms && !G.Mrows) ? g.Ms + (long long)step * g.slot : nullptr), g.ldh,
                          (last || !G.Mrows) ? nullptr : G.Mrows + (long long)step * G.mrow_slot);
        stamp();  // 6, 8, ...: update epilogue + message / aggregate
    }

    // ================= K5: out = tau(W_o [V || Mv] + b_o) on the tile's atoms =================
    {
        f32x4 acc[RT_A][WL];
        zero_acc(RA{}, acc);
        const ColConst cc = col_consts(G.WoM.inv_scale, g.b_o);
        // the first 128 columns of the tile's V rows are fetched under the Mv contraction
        const rsrc_t rV = gemm::make_rsrc(g.V + (long long)va * g.ldv, (unsigned)(na * g.ldv) * 4u);
        const rsrc_t rnull = gemm::make_rsrc(g.W_o, 0);
        unsigned rov[JR * RT_A];
#pragma unroll
        for (int j = 0; j < JR * RT_A; ++j) rov[j] = wave + NW * j < na ? (unsigned)(wave + NW * j) * (unsigned)g.ldv * 4u : kOOB;
        u32x2 v_grp[JR * RT_A];
        ga_load(RA{}, F_{}, 0, g.d_v, 0, rV, rnull, rov, rov, v_grp);
        // Mv part first (A = T16 rows 0..atoms-1), then the V part in the scale of its own groups
        contract(RA{}, acc, T16, TS, (N + BK - 1) / BK, 0, G.WoM);
        stamp();  // finalize: Mv part
        float sV = sA;
        for (int grp = 0; grp * 4 < G.WoV.nc; ++grp) {
            if (grp > 0) ga_load(RA{}, F_{}, grp, g.d_v, 0, rV, rnull, rov, rov, v_grp);
            sV = ga_stage(RA{}, acc, v_grp, sV);
            if (grp == 0) stamp();  // finalize: V staged
            const int ncg = G.WoV.nc - grp * 4 < 4 ? G.WoV.nc - grp * 4 : 4;
            contract(RA{}, acc, Ag, TSG, ncg, grp * 4, G.WoV);
        }
        stamp();  // finalize: V part
        unscale(RA{}, acc, rcp_pow2_exact(sV), cc);
        act_frags(RA{}, F_{}, acc, acc);
        dropout_frags(RA{}, acc, (unsigned)(T_steps - 1), va, false);  // dropout(tau(W_o [...])): base.py:182
        if (maxbits[5]) {  // (uniform; written before the first barrier of the kernel)
#pragma unroll
            for (int rt = 0; rt < RT_A; ++rt)
#pragma unroll
                for (int ct = 0; ct < WL; ++ct) acc[rt][ct] = f32x4{__int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000)};
        }
        if ((pf_word[0] ^ pf_word[1] ^ pf_word[2] ^ pf_word[3]) == 0x7fdead01u) maxbits[7] = 1u;   // (the L2 warm-up's loads end here: a use the compiler must wait for, of a word nothing reads)
        __syncthreads();  // (the fp32 tile may overlay the V operand tile)
        frag_to_tile(RA{}, acc);
        __syncthreads();
        tile_to_global(g.out, va, g.ldout, na);
        stamp();  // output stored
        if constexpr (KEEP) {
            if (G.agg_Hm) {   // (uniform) the tile's molecules: batch[va] .. batch[vb - 1]; T still holds their rows
                const long long m0 = G.agg_batch[va], m1 = G.agg_batch[vb - 1];
                if (m0 >= 0 && m1 < G.agg_n_mols && m1 - m0 < BA) {
                    const int nm = (int)(m1 - m0) + 1;
                    for (int it = tid; it < nm * QN; it += KT) {
                        const int ml = it / QN, q = it - ml * QN;
                        if (q >= qn) continue;
                        const long long m = m0 + ml;
                        const int f = G.agg_bounds[m] - va, e = G.agg_bounds[G.agg_n_mols + m] - va;
                        if (f < 0 || e > na || e <= f) continue;   // (not inside this tile, or no atoms: the head's own aggregation)
                        float4 sacc = *reinterpret_cast<const float4*>(T + f * LDC + 4 * q);   // include_self=False: the first addend is copied
                        for (int a = f + 1; a < e; ++a) {
                            const float4 x = *reinterpret_cast<const float4*>(T + a * LDC + 4 * q);
                            sacc.x += x.x; sacc.y += x.y; sacc.z += x.z; sacc.w += x.w;
                        }
                        if (G.agg_mode == DMPNN_MOLAGG_MEAN) { const float n = (float)(e - f); sacc.x /= n; sacc.y /= n; sacc.z /= n; sacc.w /= n; }
                        if (G.agg_mode == DMPNN_MOLAGG_NORM) { sacc.x /= G.agg_norm; sacc.y /= G.agg_norm; sacc.z /= G.agg_norm; sacc.w /= G.agg_norm; }
                        *reinterpret_cast<float4*>(G.agg_Hm + m * G.agg_ld + 4 * q) = sacc;
                        if (q == 0) G.agg_done[m] = 1;
                    }
                }
            }
        }
    }
    };  // body
    if constexpr (NW == 4) {
        body(std::integral_constant<int, WN>{});
    } else {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < 4) body(std::integral_constant<int, 3>{});
        else body(std::integral_constant<int, 2>{});
    }
}

template <int WN, bool SA, bool KEEP, int NW = 4, bool LP = false>
int launch_mega16(const Mega16K& g, int n_tiles, hipStream_t s);

#define DMPNN_DEFINE_MEGA16(WN, SA, KEEP) DMPNN_DEFINE_MEGA16_NW(WN, SA, KEEP, 4)
#define DMPNN_DEFINE_MEGA16_NW(WN, SA, KEEP, NW) DMPNN_DEFINE_MEGA16_LP(WN, SA, KEEP, NW, false)
#define DMPNN_DEFINE_MEGA16_LP(WN, SA, KEEP, NW, LP)                                                       \
    template <>                                                                                            \
    int launch_mega16<WN, SA, KEEP, NW, LP>(const Mega16K& g, int n_tiles, hipStream_t s) {                \
        const size_t lds = g.atom_de ? lds_bytes_atom<WN>() : lds_bytes<WN>();                             \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mpnn_tile16<WN, SA, KEEP, NW, LP>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_atom<WN>()); \
            if (e != hipSuccess) {                                                                         \
                set_error("hipFuncSetAttribute(k_mpnn_tile16<%d>, %zu B LDS): %s", WN, lds, hipGetErrorString(e)); \
                return DMPNN_EHIP;                                                                         \
            }                                                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL((k_mpnn_tile16<WN, SA, KEEP, NW, LP>), dim3((unsigned)n_tiles), dim3(64 * NW), lds, s, g);   \
        DMPNN_CHECK_LAUNCH("k_mpnn_tile16");                                                               \
        return DMPNN_OK;                                                                                   \
    }

}  // namespace mega16
}  // namespace dmpnn
