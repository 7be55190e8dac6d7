// extern "C" surface of libdmpnn (see include/dmpnn.h) and the forward driver that chains the
// row kernels exactly as chemprop/nn/message_passing/base.py:196-212 chains its ATen ops.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "dmpnn_common.hpp"
#include "dmpnn_mega16_impl.hpp"   // (mega16::SplitArgs: the argument block of the weight pre-split)

namespace dmpnn {

static thread_local char g_err[512] = "";
static thread_local int g_launches = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// DMPNN_DEBUG_LDS_POISON=1 (debugging aid, round 6): after EVERY launch of the library the whole LDS of every CU is filled with a NaN
// pattern (fp32 NaN = two f16 NaNs), on the legacy default stream — the stream the tests run on.  LDS is not cleared between kernels: a
// kernel that reads LDS it never wrote multiplies, on a warm box, the previous kernel's finite leftovers (by a zero weight, typically:
// unnoticed) and on a cold box whatever the last tenant left — round 6 met one such NaN on a fresh box (k_head_rows: three columns of
// a 16 x 4 tile never zeroed) and found it with this switch; `pytest -m gpu` under it is the regression test.
static __global__ __launch_bounds__(1024) void k_debug_lds_poison(unsigned pat, unsigned* sink) {
    extern __shared__ unsigned lds_words[];
    constexpr int n = (160 * 1024 - 64) / 4;
    for (int i = threadIdx.x; i < n; i += 1024) lds_words[i] = pat;
    __syncthreads();
    if (sink && lds_words[(threadIdx.x * 7) % n] == 0x12345u) sink[0] = 1;   // (keeps the stores)
}
static int g_lds_poison = -1;   // -1: not decided yet (the environment), 0 off, 1 on (dmpnn_debug_lds_poison)
static void debug_lds_poison() {
    if (g_lds_poison < 0) {
        const char* e = getenv("DMPNN_DEBUG_LDS_POISON");
        g_lds_poison = (e && e[0] == '1') ? 1 : 0;
    }
    if (!g_lds_poison) return;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_debug_lds_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) == hipSuccess;
    if (!attr) return;
    // (one workgroup per CU at this LDS size; three rounds' worth so that every CU is hit whatever the dispatch order)
    hipLaunchKernelGGL(k_debug_lds_poison, dim3(768), dim3(1024), 160 * 1024 - 64, static_cast<hipStream_t>(nullptr), 0x7FC07FC0u, static_cast<unsigned*>(nullptr));
    (void)hipGetLastError();
}
// DMPNN_TRACE=1: print every kernel launch and synchronise after it, so a device fault is
// attributed to the launch that caused it (debugging aid; never set in production).
void count_launch(const char* name) {
    ++g_launches;
    debug_lds_poison();
    static const bool trace = [] {
        const char* e = getenv("DMPNN_TRACE");
        return e && e[0] == '1';
    }();
    if (trace) {
        fprintf(stderr, "[dmpnn] launch %d: %s ... ", g_launches, name);
        fflush(stderr);
        const hipError_t e = hipDeviceSynchronize();
        fprintf(stderr, "%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
        fflush(stderr);
    }
}

static int check_graph_sizes(int64_t nV, int64_t nE) {
    DMPNN_CHECK_ARG(nV >= 0 && nE >= 0, "negative graph size (n_atoms=%lld n_edges=%lld)", (long long)nV, (long long)nE);
    DMPNN_CHECK_ARG(nV < (int64_t(1) << 31) - 64 && nE < (int64_t(1) << 31) - 64,
                    "graph too large for int32 indices (n_atoms=%lld n_edges=%lld)", (long long)nV, (long long)nE);
    return DMPNN_OK;
}

}  // namespace dmpnn

using namespace dmpnn;
namespace dmpnn { extern thread_local long long* g_debug_stamps; }

extern "C" {

int dmpnn_version(void) { return DMPNN_ABI_VERSION; }
void dmpnn_debug_lds_poison(int on) { dmpnn::g_lds_poison = on ? 1 : 0; }
int dmpnn_dropout_keep(uint64_t seed, int32_t site, int64_t row, int64_t col, float p) {
    if (!(p > 0.f && p < 1.f)) return p <= 0.f ? 1 : 0;
    return drop_hash((unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32), (unsigned)site, (unsigned)row, (unsigned)col) >= drop_threshold(p) ? 1 : 0;
}
int dmpnn_debug_timestamps(void* device_buf) {
    g_debug_stamps = static_cast<long long*>(device_buf);
    return DMPNN_OK;
}
const char* dmpnn_last_error_string(void) { return g_err; }
int dmpnn_last_launch_count(void) { return g_launches; }

size_t dmpnn_plan_bytes(int64_t n_atoms, int64_t n_edges) {
    if (n_atoms < 0 || n_edges < 0) return 0;
    return (size_t)plan_layout(n_atoms, n_edges).words * sizeof(int);
}

int dmpnn_plan_layout(int64_t n_atoms, int64_t n_edges, int64_t off[DMPNN_PLAN_NOFFSETS]) {
    DMPNN_CHECK_ARG(off != nullptr, "plan_layout: null output");
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    const PlanLayout L = plan_layout(n_atoms, n_edges);
    off[0] = L.src; off[1] = L.dst; off[2] = L.rev; off[3] = L.row_ptr; off[4] = L.perm;
    off[5] = L.inv; off[6] = L.srcp; off[7] = L.dstp; off[8] = L.revp;
    off[9] = L.tile_row; off[10] = L.tile_atom; off[11] = L.max_tiles;
    off[12] = L.mtile_row; off[13] = L.mtile_atom; off[14] = L.max_mtiles;
    return DMPNN_OK;
}

static int prepare_impl(const int64_t* edge_index, const int64_t* rev, int64_t n_atoms, int64_t n_edges, void* plan,
                        size_t plan_bytes, int light, void* stream) {
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan != nullptr, "prepare: null plan");
    DMPNN_CHECK_ARG(n_edges == 0 || (edge_index && rev), "prepare: null index arrays");
    if (plan_bytes < dmpnn_plan_bytes(n_atoms, n_edges)) {
        set_error("prepare: plan buffer too small (%zu < %zu bytes)", plan_bytes, dmpnn_plan_bytes(n_atoms, n_edges));
        return DMPNN_ENOSPC;
    }
    DMPNN_CHECK_ARG(aligned16(plan), "prepare: plan must be 16-byte aligned");
    return launch_prepare(edge_index, rev, n_atoms, n_edges, static_cast<int*>(plan), light, static_cast<hipStream_t>(stream));
}
int dmpnn_prepare(const int64_t* edge_index, const int64_t* rev, int64_t n_atoms, int64_t n_edges, void* plan,
                  size_t plan_bytes, void* stream) {
    return prepare_impl(edge_index, rev, n_atoms, n_edges, plan, plan_bytes, 0, stream);
}
int dmpnn_prepare_light(const int64_t* edge_index, const int64_t* rev, int64_t n_atoms, int64_t n_edges, void* plan,
                        size_t plan_bytes, void* stream) {
    return prepare_impl(edge_index, rev, n_atoms, n_edges, plan, plan_bytes, 1, stream);
}

int dmpnn_prepare_tiles(const int64_t* edge_index, const int64_t* rev, const int64_t* batch, int64_t n_atoms, int64_t n_edges,
                        void* plan, size_t plan_bytes, void* stream) {
    if (batch && n_atoms > 0 && small_plan_fits(n_atoms, n_edges)) {
        DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
        DMPNN_CHECK_ARG(plan != nullptr && aligned16(plan), "prepare_tiles: plan must be a 16-byte aligned buffer");
        DMPNN_CHECK_ARG(n_edges == 0 || edge_index, "prepare_tiles: null edge_index");
        if (plan_bytes < dmpnn_plan_bytes(n_atoms, n_edges)) {
            set_error("prepare_tiles: plan buffer too small (%zu < %zu bytes)", plan_bytes, dmpnn_plan_bytes(n_atoms, n_edges));
            return DMPNN_ENOSPC;
        }
        return launch_prepare_tiles_batch(edge_index, batch, n_atoms, n_edges, static_cast<int*>(plan), static_cast<hipStream_t>(stream));
    }
    if (batch && dmpnn_tile_plan_any_size(n_atoms, n_edges) && !small_plan_fits(n_atoms, n_edges)) {
        // a batch beyond the single-workgroup plan: the same tables from three multi-workgroup launches
        DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
        DMPNN_CHECK_ARG(plan != nullptr && aligned16(plan), "prepare_tiles: plan must be a 16-byte aligned buffer");
        DMPNN_CHECK_ARG(n_edges == 0 || edge_index, "prepare_tiles: null edge_index");
        if (plan_bytes < dmpnn_plan_bytes(n_atoms, n_edges)) {
            set_error("prepare_tiles: plan buffer too small (%zu < %zu bytes)", plan_bytes, dmpnn_plan_bytes(n_atoms, n_edges));
            return DMPNN_ENOSPC;
        }
        return launch_prepare_tiles_large(edge_index, batch, n_atoms, n_edges, static_cast<int*>(plan), static_cast<hipStream_t>(stream));
    }
    return prepare_impl(edge_index, rev, n_atoms, n_edges, plan, plan_bytes, 2, stream);
}

}  // extern "C"
namespace dmpnn {
int prepare_tiles_and_bounds(const int64_t* edge_index, const int64_t* rev, const int64_t* batch, int64_t n_atoms, int64_t n_edges, void* plan,
                             size_t plan_bytes, int* mol_bounds, int64_t n_mols, void* stream, bool* wrote_bounds,
                             const dmpnn_fwd_args* split_for, bool* did_split) {
    if (wrote_bounds) *wrote_bounds = false;
    if (did_split) *did_split = false;
    if (mol_bounds && (n_mols <= 0 || n_mols >= (1 << 30))) mol_bounds = nullptr;
    // the single-workgroup planner from the batch vector: the aggregation's bounds on the side, the weight pre-split in the same launch
    if ((mol_bounds || split_for) && batch && n_atoms > 0 && small_plan_fits(n_atoms, n_edges) && check_graph_sizes(n_atoms, n_edges) == DMPNN_OK &&
        plan != nullptr && aligned16(plan) && (n_edges == 0 || edge_index) && plan_bytes >= dmpnn_plan_bytes(n_atoms, n_edges)) {
        DMPNN_TRY(launch_prepare_tiles_batch(edge_index, batch, n_atoms, n_edges, static_cast<int*>(plan), static_cast<hipStream_t>(stream), mol_bounds,
                                             mol_bounds ? n_mols : 0, split_for, did_split));
        if (wrote_bounds) *wrote_bounds = mol_bounds != nullptr;
        return DMPNN_OK;
    }
    return dmpnn_prepare_tiles(edge_index, rev, batch, n_atoms, n_edges, plan, plan_bytes, stream);
}
}  // namespace dmpnn
extern "C" {

int dmpnn_tile_plan_any_size(int64_t n_atoms, int64_t n_edges) { return tiles_large_fits(n_atoms, n_edges) ? 1 : 0; }
int dmpnn_full_plan_keeps_tiles(int64_t n_atoms, int64_t n_edges) {
    if (n_atoms <= 0 || n_edges <= 0) return 0;
    if (small_plan_fits(n_atoms, n_edges)) return 1;
    return (tiles_large_fits(n_atoms, n_edges) && prepare_can_keep_mtiles(n_atoms, n_edges)) ? 1 : 0;
}

int dmpnn_prepare_with_batch(const int64_t* edge_index, const int64_t* rev, const int64_t* batch, int64_t n_atoms, int64_t n_edges,
                             void* plan, size_t plan_bytes, void* stream) {
    if (!batch || n_atoms <= 0 || n_edges <= 0 || small_plan_fits(n_atoms, n_edges) || !tiles_large_fits(n_atoms, n_edges) ||
        !prepare_can_keep_mtiles(n_atoms, n_edges))
        return prepare_impl(edge_index, rev, n_atoms, n_edges, plan, plan_bytes, 0, stream);
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan != nullptr && aligned16(plan), "prepare_with_batch: plan must be a 16-byte aligned buffer");
    DMPNN_CHECK_ARG(edge_index && rev, "prepare_with_batch: null index arrays");
    if (plan_bytes < dmpnn_plan_bytes(n_atoms, n_edges)) {
        set_error("prepare_with_batch: plan buffer too small (%zu < %zu bytes)", plan_bytes, dmpnn_plan_bytes(n_atoms, n_edges));
        return DMPNN_ENOSPC;
    }
    // the tiles first (their scratch is the arrays the full plan fills afterwards), then the full plan around them
    DMPNN_TRY(launch_prepare_tiles_large(edge_index, batch, n_atoms, n_edges, static_cast<int*>(plan), static_cast<hipStream_t>(stream)));
    return launch_prepare(edge_index, rev, n_atoms, n_edges, static_cast<int*>(plan), 0, static_cast<hipStream_t>(stream), true);
}

int dmpnn_message_fwd(const void* plan, int64_t n_atoms, int64_t n_edges, int64_t d_h, const float* Hin,
                      int64_t ld_in, float* M, int64_t ld_m, int act_on_load, float act_slope,
                      const float* act_slope_ptr, unsigned flags, void* stream) {
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan && d_h >= 0 && ld_in >= d_h && ld_m >= d_h, "message_fwd: bad arguments");
    DMPNN_CHECK_ARG(n_edges == 0 || (Hin && M), "message_fwd: null tensor");
    DMPNN_CHECK_ARG(act_on_load != DMPNN_ACT_PRELU || act_slope_ptr, "message_fwd: PReLU needs act_slope_ptr");
    return launch_message(plan_view(plan, n_atoms, n_edges), n_atoms, n_edges, d_h, Hin, ld_in, M, ld_m,
                          act_on_load, act_slope, act_slope_ptr, flags, static_cast<hipStream_t>(stream));
}

int dmpnn_aggregate_fwd(const void* plan, int64_t n_atoms, int64_t n_edges, int64_t d_h, const float* Hin,
                        int64_t ld_in, float* Mv, int64_t ld_mv, int act_on_load, float act_slope,
                        const float* act_slope_ptr, void* stream) {
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan && d_h >= 0 && ld_in >= d_h && ld_mv >= d_h, "aggregate_fwd: bad arguments");
    DMPNN_CHECK_ARG(n_atoms == 0 || Mv, "aggregate_fwd: null output");
    DMPNN_CHECK_ARG(n_edges == 0 || Hin, "aggregate_fwd: null input");
    DMPNN_CHECK_ARG(act_on_load != DMPNN_ACT_PRELU || act_slope_ptr, "aggregate_fwd: PReLU needs act_slope_ptr");
    return launch_aggregate(plan_view(plan, n_atoms, n_edges), n_atoms, n_edges, d_h, Hin, ld_in, Mv, ld_mv,
                            act_on_load, act_slope, act_slope_ptr, static_cast<hipStream_t>(stream));
}

int dmpnn_linear_fwd(const dmpnn_gemm_args* a, void* stream) {
    DMPNN_CHECK_ARG(a != nullptr, "linear_fwd: null args");
    DMPNN_CHECK_ARG(a->act != DMPNN_ACT_PRELU || a->act_slope_ptr, "linear_fwd: PReLU needs act_slope_ptr");
    return launch_linear(*a, static_cast<hipStream_t>(stream));
}

int dmpnn_update_fwd(const void* plan, int64_t n_atoms, int64_t n_edges, int64_t d_h, const float* M, int64_t ld_m,
                     const float* H0, int64_t ld_h0, const float* W_h, const float* b_h, float* H_out, int64_t ld_hout,
                     float* M_next, int64_t ld_mnext, float* Mv, int64_t ld_mv, int act, float act_slope,
                     const float* act_slope_ptr, void* stream) {
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan && d_h > 0, "update_fwd: bad arguments");
    DMPNN_CHECK_ARG(n_edges == 0 || (M && H0 && W_h), "update_fwd: null tensor");
    DMPNN_CHECK_ARG(M_next || Mv || H_out, "update_fwd: no output");
    DMPNN_CHECK_ARG(act != DMPNN_ACT_PRELU || act_slope_ptr, "update_fwd: PReLU needs act_slope_ptr");
    if (n_edges == 0) {
        if (Mv && n_atoms > 0) {
            for (int64_t v = 0; v < n_atoms; ++v)
                if (hipMemsetAsync(Mv + v * ld_mv, 0, (size_t)d_h * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess) {
                    set_error("update_fwd: memset failed");
                    return DMPNN_EHIP;
                }
        }
        return DMPNN_OK;
    }
    const int* plan_i = static_cast<const int*>(plan);
    const PlanLayout L = plan_layout(n_atoms, n_edges);
    dmpnn_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.M = n_edges; g.N = d_h; g.K1 = d_h;
    g.A1 = M; g.lda1 = ld_m;
    g.W = W_h; g.ldw = d_h; g.bias = b_h;
    g.Cadd = H0; g.ldcadd = ld_h0;
    g.C = H_out; g.ldc = ld_hout;
    g.act = act; g.act_slope = act_slope; g.act_slope_ptr = act_slope_ptr;
    GemmExtra x;
    memset(&x, 0, sizeof(x));
    x.seg = true;
    x.tile_row = plan_i + L.tile_row; x.tile_atom = plan_i + L.tile_atom; x.n_tiles = (int)L.max_tiles;
    x.row_ptr = plan_i + L.row_ptr; x.revp = plan_i + L.revp;
    x.Mout = M_next; x.ldm = ld_mnext; x.Sout = Mv; x.lds = ld_mv;
    return launch_linear_ex(g, x, static_cast<hipStream_t>(stream));
}

// per-step routes with DMPNN_F_SPLIT16: W_i | W_h | W_o (| W_d) pre-split one after the other
static size_t steps16_wsplit_bytes(const dmpnn_fwd_args& a) {
    const int64_t h = a.d_h;
    size_t n = linear16_wsplit_bytes(h, a.d_v + a.d_e) + linear16_wsplit_bytes(h, h) + linear16_wsplit_bytes(h, a.d_v + h);
    n += linear16_wsplit_bytes(h + a.d_vd, h + a.d_vd) * ((a.W_d && a.d_vd > 0) ? 1 : 0);
    n += linear16_wsplit_bytes(h, h) + linear16_wsplit_bytes(h, a.d_v);  // W_o[:, d_v:] | W_o[:, :d_v] on their own (the finalize on the step kernel)
    return n;
}

size_t dmpnn_forward_wsplit_bytes(const dmpnn_fwd_args* a) {
    if (!a || a->d_h <= 0 || a->d_v <= 0 || a->d_e < 0) return 0;
    return (a->flags & DMPNN_F_MEGA) ? mega16_wsplit_bytes(*a) : steps16_wsplit_bytes(*a);
}

// one bit per element of H0 and of every kept H^(t): 256 words of 8 bytes per tile (64 per wave, RT_E WN 4 <= 60 of them used)
static bool keep_bits_apply(const dmpnn_fwd_args& a) {
    const unsigned need = DMPNN_F_TILE_PLAN | DMPNN_F_KEEP | DMPNN_F_MEGA | DMPNN_F_SPLIT16 | DMPNN_F_FUSED;
    return (a.flags & need) == need && (a.act == DMPNN_ACT_NONE || a.act == DMPNN_ACT_RELU || a.act == DMPNN_ACT_LEAKYRELU) &&
           !(a.dropout_p > 0.f) && !a.W_d && a.n_atoms > 0 && a.n_edges > 0;
}
size_t dmpnn_forward_keep_bits_bytes(const dmpnn_fwd_args* a) {
    if (!a) return 0;
    // the per-step fused route on the f16 pipe (DMPNN_F_FUSED | DMPNN_F_SPLIT16 without DMPNN_F_MEGA): its LEAN training forward
    if ((a->flags & (DMPNN_F_FUSED | DMPNN_F_SPLIT16)) == (DMPNN_F_FUSED | DMPNN_F_SPLIT16) && !(a->flags & DMPNN_F_MEGA)) return fused16_lean_bits_bytes(*a);
    if (!keep_bits_apply(*a)) return 0;
    return (size_t)a->depth * (size_t)plan_layout(a->n_atoms, a->n_edges).max_mtiles * 256u * 8u;
}

size_t dmpnn_forward_h0_bytes(const dmpnn_fwd_args* a) { return a ? fused16_h0q_bytes(*a) : 0; }

size_t dmpnn_forward_spill_bytes(const dmpnn_fwd_args* a) {
    if (!a || a->n_atoms < 0 || a->n_edges < 0 || a->ldh <= 0) return 0;
    return (size_t)(3 * a->n_edges + a->n_atoms) * (size_t)a->ldh * sizeof(float);
}

static bool al_ptr(const void* p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(bytes - 1)) == 0; }

int dmpnn_forward_can_fuse(const dmpnn_fwd_args* a) {
    if (!a) return 0;
    const int64_t h = a->d_h, dv = a->d_v, de = a->d_e;
    if (a->flags & DMPNN_F_UNDIRECTED) return 0;
    if (h <= 0 || h % 4 != 0 || h > 320 || a->ldh % 4 != 0) return 0;
    if (dv % 2 != 0 || de % 2 != 0 || a->ldv % 2 != 0 || a->lde % 2 != 0) return 0;
    if (!al_ptr(a->V, 8) || !al_ptr(a->E, 8) || !al_ptr(a->W_i, 8) || !al_ptr(a->W_h, 16)) return 0;
    if (!al_ptr(a->H0, 16) || !al_ptr(a->Ms, 16) || !al_ptr(a->Mv, 16) || (a->Hs && !al_ptr(a->Hs, 16))) return 0;
    if (a->n_atoms * a->ldv * 4 > 0x7FFFFFFF || a->n_edges * a->lde * 4 > 0x7FFFFFFF) return 0;
    return mega_shapes_ok(*a) ? 2 : 1;
}

// measured crossovers (MI355X): from this many directed edges on the per-step routes run their contractions on the f16 pipe /
// an inference forward that is not the tile kernel's takes the per-step fused route on the f16 pipe
static const int64_t kSteps16MinEdges = 20000, kFused16MinEdges = 2048;

int dmpnn_tile_waves(int64_t n_atoms, int64_t n_edges, int64_t d_h, int64_t n_tiles) {
    dmpnn_fwd_args t;
    memset(&t, 0, sizeof(t));
    t.n_atoms = n_atoms; t.n_edges = n_edges; t.d_h = d_h;
    return tile_waves(t, n_tiles > 0 ? (int)n_tiles : (int)plan_layout(n_atoms, n_edges).max_mtiles);
}

int dmpnn_forward_route(const dmpnn_fwd_args* a, int keep, int max_level, int plan_kind, int arith) {
    if (!a || max_level < 0 || plan_kind < 0 || plan_kind > 2 || arith < 0 || arith > 1) return -1;
    const bool undirected = (a->flags & DMPNN_F_UNDIRECTED) != 0;
    int level = undirected ? 0 : dmpnn_forward_can_fuse(a);
    if (level > max_level) level = max_level;
    // a tile plan: the whole-forward tile kernel on the f16 pipe or nothing — inference, and training without W_d (the kept tensors
    // are then in the caller's edge order: the forward is given DMPNN_F_TILE_PLAN for dmpnn_backward to know)
    if (plan_kind == 2) return (level == 2 && arith == 0 && !(keep && a->W_d)) ? DMPNN_ROUTE_MEGA16 : -1;
    if (plan_kind == 1 && keep) return -1;
    const int64_t nE = a->n_edges, h = a->d_h;
    if (!keep && !undirected && arith == 0 && nE >= kFused16MinEdges && (level == 1 || (level == 0 && h > 320 && max_level >= 1))) {
        dmpnn_fwd_args t = *a;
        t.flags &= ~(unsigned)(DMPNN_F_KEEP | DMPNN_F_STORE16);
        if (fused16_shapes_ok(t)) return DMPNN_ROUTE_FUSED16;
    }
    // training at size, molecules beyond the tile: the per-step FUSED route's lean forward + the backward step kernels (round 4:
    // ReLU-class activation, no W_d, d_h <= 320) — else the per-step general route on the f16 pipe
    if (keep && level == 1 && arith == 0 && nE >= kSteps16MinEdges && plan_kind == 0) {
        dmpnn_fwd_args t = *a;
        t.flags = (t.flags | DMPNN_F_FUSED | DMPNN_F_SPLIT16) & ~(unsigned)(DMPNN_F_MEGA | DMPNN_F_STORE16);
        if (fused16_lean_shapes(t)) return DMPNN_ROUTE_FUSED16;
    }
    if (level == 1 && arith == 0 && nE >= kSteps16MinEdges && plan_kind != 1) level = 0;  // (training at size: the per-step route on the f16 pipe)
    if (level == 2) return arith == 0 ? DMPNN_ROUTE_MEGA16 : DMPNN_ROUTE_MEGA;
    if (level == 1) return DMPNN_ROUTE_FUSED;
    if (plan_kind == 1) return -1;
    return (arith == 0 && (h > 320 || nE >= kSteps16MinEdges)) ? DMPNN_ROUTE_GENERAL16 : DMPNN_ROUTE_GENERAL;
}

int dmpnn_train_route(const dmpnn_fwd_args* a, int64_t n_mols, int32_t have, int32_t oversize, int32_t max_level, int32_t arith,
                      int32_t keep_rows, dmpnn_train_route_info* out) {
    DMPNN_CHECK_ARG(a && out && max_level >= 0 && (arith == 0 || arith == 1) && keep_rows >= -1 && keep_rows <= 1, "train_route: bad arguments");
    memset(out, 0, sizeof(*out));
    const int64_t nV = a->n_atoms, nE = a->n_edges, h = a->d_h;
    const bool undirected = (a->flags & DMPNN_F_UNDIRECTED) != 0, atom = (a->flags & DMPNN_F_ATOM) != 0;
    const bool relu_class = a->act == DMPNN_ACT_RELU || a->act == DMPNN_ACT_LEAKYRELU;
    const bool builtin = a->act == DMPNN_ACT_NONE || relu_class || a->act == DMPNN_ACT_TANH || a->act == DMPNN_ACT_ELU;   // (not PReLU: its slope trains)
    // ---- which plan ----
    bool tiles = !undirected && !a->W_d && builtin && max_level >= 2 && arith == 0 && nE > 0 && nV > 0 && oversize != 1;
    if (a->dropout_p > 0.f && !relu_class) tiles = false;           // (the mask is recovered from the sign of the kept tensors)
    if (!(h > 0 && h % 4 == 0 && h <= 320 && a->d_v % 2 == 0 && a->d_e % 2 == 0)) tiles = false;
    if (atom && !(a->d_e >= 2 && a->d_e <= 16) ) tiles = false;
    const bool small = small_plan_fits(nV, nE), table = (have & 2) != 0, large = (have & 1) != 0 && tiles_large_fits(nV, nE);
    if (!(small || table || large)) tiles = false;                  // a planner that can build it
    if (!(n_mols > 0 && nE <= 30 * n_mols && (table || large || small))) tiles = false;   // (30 directed edges per molecule: near the tile)
    out->plan_kind = tiles ? 2 : 0;
    // ---- the route on that plan ----
    int cap = max_level;
    if (oversize == 1 && cap > 1) cap = 1;
    dmpnn_fwd_args t = *a;
    if (table || large) t.flags |= DMPNN_F_LOADER_TILES;
    out->route = dmpnn_forward_route(&t, 1, cap, out->plan_kind, arith);
    if (out->route < 0 && out->plan_kind == 2) {                    // (the shapes did not reach the tile kernel after all: the full plan)
        out->plan_kind = 0;
        out->route = dmpnn_forward_route(&t, 1, cap, 0, arith);
    }
    // ---- the form of the kept tensors ----
    const int64_t n_steps = a->depth > 1 ? a->depth - 1 : 0;
    // (keep_rows is a size rule of the tile kernels' training forward — meaningful when that is the route — and answered whatever
    //  `route` says, so that a caller that already holds the route, engine.forward, can ask for it alone)
    const bool rows = keep_rows == 1 || (keep_rows < 0 && nE * n_steps >= DMPNN_KEEP_ROWS_MIN);
    out->keep_rows = (!atom && n_steps > 0 && h > 0 && h <= 320 && rows) ? 1 : 0;
    if (out->route == DMPNN_ROUTE_MEGA16) {
        out->keep_bits = (out->plan_kind == 2 && (relu_class || a->act == DMPNN_ACT_NONE) && !(a->dropout_p > 0.f)) ? 1 : 0;
    } else if (out->route == DMPNN_ROUTE_FUSED16) {
        dmpnn_fwd_args l = *a;
        l.flags = (l.flags | DMPNN_F_FUSED | DMPNN_F_SPLIT16) & ~(unsigned)(DMPNN_F_MEGA | DMPNN_F_STORE16);
        out->lean = fused16_lean_shapes(l) ? 1 : 0;
        out->keep_bits = out->lean;
    }
    return DMPNN_OK;
}

int dmpnn_forward_tiles(const dmpnn_fwd_args* a, const int64_t* batch, const int* tile_row, const int* tile_atom, int64_t n_tiles,
                        size_t plan_bytes, void* stream) {
    DMPNN_CHECK_ARG(a != nullptr && a->plan != nullptr, "forward_tiles: null args / plan");
    void* plan = const_cast<void*>(a->plan);
    bool did_split = false;
    if (tile_row && tile_atom && n_tiles > 0) {
        // the loader's table: copied / checked by ONE workgroup — with the forward's weight pre-split in the same launch where the tile
        // kernel on the f16 pipe is going to run (the arguments are checked as dmpnn_prepare_tiles_from_table checks them)
        const dmpnn::PlanLayout L = dmpnn::plan_layout(a->n_atoms, a->n_edges);
        const bool ok = a->n_atoms >= 0 && a->n_edges >= 0 && a->n_atoms < (1ll << 31) && a->n_edges < (1ll << 31) &&
                        plan_bytes >= (size_t)L.words * sizeof(int) && n_tiles <= L.max_mtiles;
        if (ok) DMPNN_TRY(launch_tiles_from_table_split(tile_row, tile_atom, n_tiles, a->n_atoms, a->n_edges, static_cast<int*>(plan),
                                                        static_cast<hipStream_t>(stream), a, &did_split));
        if (!did_split)
            DMPNN_TRY(dmpnn_prepare_tiles_from_table(tile_row, tile_atom, n_tiles, a->n_atoms, a->n_edges, plan, plan_bytes, stream));
    } else   // (from the batch vector within the single-workgroup plan: the weight pre-split rides in K0's launch)
        DMPNN_TRY(prepare_tiles_and_bounds(a->edge_index, a->rev_edge_index, batch, a->n_atoms, a->n_edges, plan, plan_bytes, nullptr, 0, stream, nullptr, a, &did_split));
    if (!did_split) return dmpnn_forward(a, stream);
    dmpnn_fwd_args f = *a;
    f.flags |= DMPNN_F_WSPLIT_READY;
    return dmpnn_forward(&f, stream);
}

int dmpnn_forward(const dmpnn_fwd_args* a, void* stream) {
    g_launches = 0;
    DMPNN_CHECK_ARG(a != nullptr, "forward: null args");
    DMPNN_TRY(check_graph_sizes(a->n_atoms, a->n_edges));
    const int64_t nV = a->n_atoms, nE = a->n_edges, h = a->d_h, dv = a->d_v, de = a->d_e;
    DMPNN_CHECK_ARG(a->plan, "forward: null plan");
    DMPNN_CHECK_ARG(h > 0 && dv > 0 && de >= 0 && a->d_vd >= 0 && a->depth >= 1, "forward: bad dimensions");
    DMPNN_CHECK_ARG(a->act >= DMPNN_ACT_RELU && a->act <= DMPNN_ACT_ELU,
                    "forward: activation %d is not built in (chain the row kernels instead)", a->act);
    DMPNN_CHECK_ARG(a->act != DMPNN_ACT_PRELU || a->act_slope_ptr, "forward: PReLU needs act_slope_ptr");
    DMPNN_CHECK_ARG(a->W_i && a->W_h && a->W_o && a->b_o, "forward: null weight");
    DMPNN_CHECK_ARG(a->ldh >= h && a->ldv >= dv && a->lde >= de, "forward: leading dimension too small");
    DMPNN_CHECK_ARG(nV == 0 || (a->V && a->Mv && a->out), "forward: null V / Mv / out");
    DMPNN_CHECK_ARG(nE == 0 || (a->E && (a->H0 || (a->flags & DMPNN_F_MEGA))), "forward: null E / H0");
    const bool has_vd = a->W_d != nullptr;
    DMPNN_CHECK_ARG(!has_vd || (a->d_vd > 0 && a->V_d && a->b_d && a->Hv && a->ldvd >= a->d_vd),
                    "forward: W_d given but V_d / b_d / Hv / d_vd missing");
    DMPNN_CHECK_ARG(a->ldout >= h + (has_vd ? a->d_vd : 0), "forward: ldout too small");
    const bool fused = a->flags & DMPNN_F_FUSED;
    DMPNN_CHECK_ARG(!(a->flags & DMPNN_F_STORE16) || (fused && (a->flags & DMPNN_F_SPLIT16) && !(a->flags & (DMPNN_F_KEEP | DMPNN_F_ATOM))),
                    "forward: DMPNN_F_STORE16 only goes with the fused routes on the f16 pipe (DMPNN_F_FUSED | DMPNN_F_SPLIT16 [| DMPNN_F_MEGA]), inference, bond messages");
    const bool lean16 = fused && (a->flags & DMPNN_F_SPLIT16) && (a->flags & DMPNN_F_KEEP) && !(a->flags & DMPNN_F_MEGA) && a->keep_bits;
    if (a->depth > 1 && nE > 0 && !(a->flags & DMPNN_F_MEGA) && !lean16) {
        DMPNN_CHECK_ARG(a->Ms && a->n_mslots >= 1, "forward: missing Ms workspace");
        DMPNN_CHECK_ARG(fused || (a->Hs && a->n_hslots >= 1), "forward: missing Hs workspace");
    }

    if (a->flags & DMPNN_F_ATOM) {
        DMPNN_CHECK_ARG((a->flags & DMPNN_F_FUSED) && (a->flags & DMPNN_F_MEGA) && (a->flags & DMPNN_F_SPLIT16) &&
                        !has_vd && de >= 1 && de <= 16 && a->dropout_p == 0.f,
                        "forward: DMPNN_F_ATOM (atom messages) runs on the whole-forward tile kernel only (DMPNN_F_FUSED | DMPNN_F_MEGA | "
                        "DMPNN_F_SPLIT16, 1 <= d_e <= 16, no W_d, no dropout inside the kernels) — chain the row kernels otherwise");
        // a TRAINING forward (round 4): the bond-feature half of the messages is kept for W_h's gradient — depth - 1 slots of [n_edges][16] in `msplit`
        DMPNN_CHECK_ARG(!(a->flags & DMPNN_F_KEEP) || (de % 2 == 0 && dv % 2 == 0 && h % 2 == 0 &&
                        (a->depth < 2 || nE == 0 || (a->msplit && aligned16(a->msplit) && a->msplit_bytes >= (size_t)(a->depth - 1) * (size_t)nE * 16 * sizeof(float)))),
                        "forward: DMPNN_F_ATOM | DMPNN_F_KEEP needs even d_v / d_e / d_h and `msplit` >= (depth - 1) * n_edges * 64 bytes, 16-byte aligned");
    }
    if (a->dropout_p != 0.f) {
        DMPNN_CHECK_ARG(a->dropout_p > 0.f && a->dropout_p < 1.f, "forward: dropout_p must lie in [0, 1)");
        const bool tile_train = (a->flags & DMPNN_F_MEGA) && (a->flags & DMPNN_F_SPLIT16) && (a->flags & DMPNN_F_KEEP);
        DMPNN_CHECK_ARG(tile_train && !has_vd && (a->act == DMPNN_ACT_RELU || a->act == DMPNN_ACT_LEAKYRELU),
                        "forward: dropout inside the kernels needs the training forward of the tile kernel (DMPNN_F_MEGA | DMPNN_F_SPLIT16 | "
                        "DMPNN_F_KEEP), a ReLU-class activation and no W_d — run dropout between the row kernels otherwise");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const PlanView pv = plan_view(a->plan, nV, nE);
    const int64_t slot = nE * a->ldh;
    const int* plan_i = static_cast<const int*>(a->plan);

    if (fused && (a->flags & DMPNN_F_SPLIT16) && !(a->flags & DMPNN_F_MEGA)) {
        // ---- per-step fused route on the f16 pipe: split message rows between the steps (dmpnn_step16_impl.hpp) ----
        DMPNN_CHECK_ARG(fused16_shapes_ok(*a), "forward: DMPNN_F_FUSED | DMPNN_F_SPLIT16 given but the shapes do not allow it "
                        "(directed, d_h %% 4 == 0, d_h <= 640, even d_v / d_e; with DMPNN_F_KEEP: depth - 1 kept H / M slots and `msplit`)");
        DMPNN_CHECK_ARG(nE == 0 || (a->H0 && ((a->flags & DMPNN_F_KEEP) ? (a->Ms || a->depth == 1 || lean16) : (a->Ms && a->n_mslots >= 2))),
                        "forward(fused16): H0 and two split message slots (training: `msplit` + the kept fp32 slots) are required");
        DMPNN_CHECK_ARG(a->wsplit && a->wsplit_bytes >= steps16_wsplit_bytes(*a), "forward(fused16): wsplit workspace missing or too small");
        SplitWView w[6];
        unsigned char* wp = static_cast<unsigned char*>(a->wsplit);
        const bool ready = (a->flags & DMPNN_F_WSPLIT_READY) != 0;
        // W_i | W_h | W_o | W_d | W_o[:, d_v:] | W_o[:, :d_v]   (the last two: column blocks of W_o, leading dimension d_v + h)
        const float* Ws[6] = {a->W_i, a->W_h, a->W_o, has_vd ? a->W_d : nullptr, a->W_o + dv, a->W_o};
        const int64_t Ns[6] = {h, h, h, h + a->d_vd, h, h}, Ks[6] = {dv + de, h, dv + h, h + a->d_vd, h, dv};
        const int64_t Ls[6] = {dv + de, h, dv + h, h + a->d_vd, dv + h, dv + h};
        SplitWJob jobs[6];
        SplitWView views[6];
        int idx[6], nj = 0;
        for (int i = 0; i < 6; ++i) {
            if (!Ws[i]) continue;
            if (ready) w[i] = split_weights_view_of(wp, Ns[i], Ks[i]);
            else { jobs[nj] = SplitWJob{Ws[i], Ls[i], Ns[i], Ks[i], 0, wp}; idx[nj++] = i; }
            wp += linear16_wsplit_bytes(Ns[i], Ks[i]);
        }
        mega16::SplitArgs sp;   // (one launch for all of them — or none: it rides in the forward's first launch, k_split_rows)
        const bool pend = nj > 0 && split_weights_args(jobs, nj, views, &sp);
        for (int k = 0; k < nj; ++k) w[idx[k]] = views[k];
        DMPNN_TRY(launch_fused16_forward(*a, w, has_vd ? a->Hv : a->out, has_vd ? a->ldh : a->ldout, s, pend ? &sp : nullptr));
        if (has_vd) {
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nV; g.N = h + a->d_vd; g.K1 = h; g.K2 = a->d_vd;
            g.A1 = a->Hv; g.lda1 = a->ldh;
            g.A2 = a->V_d; g.lda2 = a->ldvd;
            g.W = a->W_d; g.ldw = h + a->d_vd; g.bias = a->b_d;
            g.C = a->out; g.ldc = a->ldout; g.act = DMPNN_ACT_NONE;
            if (linear16_ok(g)) DMPNN_TRY(launch_linear16_view(g, w[3], nullptr, 0, s));
            else DMPNN_TRY(launch_linear(g, s));
        }
        return DMPNN_OK;
    }
    if (fused) {
        // ---- fused route: edge tensors in CSR-row order, segment sums in the contraction epilogues ----
        DMPNN_CHECK_ARG(dmpnn_forward_can_fuse(a), "forward: DMPNN_F_FUSED given but the shapes / alignment do not allow it "
                        "(d_h %% 4, d_h <= 320, even d_v / d_e, directed); call dmpnn_forward_can_fuse first");
        const PlanLayout L = plan_layout(nV, nE);
        DMPNN_CHECK_ARG((a->flags & DMPNN_F_MEGA) || a->depth <= 2 || a->n_mslots >= 2,
                        "forward(fused): depth > 2 needs at least two message slots");
        if (a->flags & DMPNN_F_MEGA) {
            // ---- whole forward of every tile of whole molecules in one launch ----
            DMPNN_CHECK_ARG(mega_shapes_ok(*a), "forward: DMPNN_F_MEGA given but the shapes do not allow it");
            DMPNN_CHECK_ARG(!a->keep_bits || (keep_bits_apply(*a) && aligned16(a->keep_bits) && a->keep_bits_bytes >= dmpnn_forward_keep_bits_bytes(a)),
                            "forward: keep_bits needs DMPNN_F_TILE_PLAN | DMPNN_F_KEEP on the tile kernel (f16 pipe), a ReLU-class activation, no "
                            "dropout, no W_d, and dmpnn_forward_keep_bits_bytes() bytes");
            DMPNN_CHECK_ARG(!(a->flags & DMPNN_F_KEEP) || a->depth == 1 || nE == 0 ||
                            (a->Hs && a->n_hslots >= a->depth - 1 && a->n_mslots >= a->depth - 1),
                            "forward(mega, keep): needs depth-1 H and M slots");
            if (nV > 0) {
                if (a->flags & DMPNN_F_SPLIT16) DMPNN_TRY(launch_mega16_forward(*a, has_vd ? a->Hv : a->out, has_vd ? a->ldh : a->ldout, s));
                else DMPNN_TRY(launch_mega_forward(*a, has_vd ? a->Hv : a->out, has_vd ? a->ldh : a->ldout, s));
            }
            if (has_vd) {
                dmpnn_gemm_args g;
                memset(&g, 0, sizeof(g));
                g.M = nV; g.N = h + a->d_vd; g.K1 = h; g.K2 = a->d_vd;
                g.A1 = a->Hv; g.lda1 = a->ldh;
                g.A2 = a->V_d; g.lda2 = a->ldvd;
                g.W = a->W_d; g.ldw = h + a->d_vd; g.bias = a->b_d;
                g.C = a->out; g.ldc = a->ldout; g.act = DMPNN_ACT_NONE;
                DMPNN_TRY(launch_linear(g, s));
            }
            return DMPNN_OK;
        }
        GemmExtra x;
        memset(&x, 0, sizeof(x));
        x.seg = true;
        x.tile_row = plan_i + L.tile_row; x.tile_atom = plan_i + L.tile_atom; x.n_tiles = nE > 0 ? (int)L.max_tiles : 0;
        x.row_ptr = plan_i + L.row_ptr; x.revp = plan_i + L.revp;
        const int T = a->depth;
        if (nE == 0 && nV > 0) {
            hipError_t e = hipMemsetAsync(a->Mv, 0, (size_t)nV * a->ldh * sizeof(float), s);
            if (e != hipSuccess) { set_error("forward: memset failed: %s", hipGetErrorString(e)); return DMPNN_EHIP; }
        }
        if (nE > 0) {
            // K1 + tau + first message (or the final aggregate when depth == 1)
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nE; g.N = h; g.K1 = dv; g.K2 = de;
            g.A1 = a->V; g.lda1 = a->ldv; g.gather1 = plan_i + L.srcp; g.gather1_rows = nV;
            g.A2 = a->E; g.lda2 = a->lde;
            g.W = a->W_i; g.ldw = dv + de; g.bias = a->b_i;
            g.Zpre = a->H0; g.ldz = a->ldh;  // pre-activation: the residual of every later step
            g.act = a->act; g.act_slope = a->act_slope; g.act_slope_ptr = a->act_slope_ptr;
            GemmExtra xi = x;
            xi.gather2 = plan_i + L.perm; xi.gather2_rows = nE;
            if (T > 1) { xi.Mout = a->Ms; xi.ldm = a->ldh; }
            else { xi.Sout = a->Mv; xi.lds = a->ldh; }
            DMPNN_TRY(launch_linear_ex(g, xi, s));
            for (int t = 1; t < T; ++t) {
                // K3 + next message / final aggregate
                memset(&g, 0, sizeof(g));
                g.M = nE; g.N = h; g.K1 = h;
                g.A1 = a->Ms + ((t - 1) % a->n_mslots) * slot; g.lda1 = a->ldh;
                g.W = a->W_h; g.ldw = h; g.bias = a->b_h;
                g.Cadd = a->H0; g.ldcadd = a->ldh;
                g.C = a->Hs ? a->Hs + ((t - 1) % a->n_hslots) * slot : nullptr; g.ldc = a->ldh;
                g.act = a->act; g.act_slope = a->act_slope; g.act_slope_ptr = a->act_slope_ptr;
                GemmExtra xu = x;
                if (t < T - 1) { xu.Mout = a->Ms + (t % a->n_mslots) * slot; xu.ldm = a->ldh; }
                else { xu.Sout = a->Mv; xu.lds = a->ldh; }
                DMPNN_TRY(launch_linear_ex(g, xu, s));
            }
        }
        // K5 finalize (atom rows: plain epilogue); a graph the fused tiling cannot hold poisons the output
        GemmExtra xf;
        memset(&xf, 0, sizeof(xf));
        xf.poison_flags = plan_i + DMPNN_HDR_FLAGS; xf.poison_mask = kPlanNoFuse;
        {
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nV; g.N = h; g.K1 = dv; g.K2 = h;
            g.A1 = a->V; g.lda1 = a->ldv;
            g.A2 = a->Mv; g.lda2 = a->ldh;
            g.W = a->W_o; g.ldw = dv + h; g.bias = a->b_o;
            g.C = has_vd ? a->Hv : a->out; g.ldc = has_vd ? a->ldh : a->ldout;
            g.act = a->act; g.act_slope = a->act_slope; g.act_slope_ptr = a->act_slope_ptr;
            DMPNN_TRY(launch_linear_ex(g, xf, s));
        }
        if (has_vd) {
            dmpnn_gemm_args g;
            memset(&g, 0, sizeof(g));
            g.M = nV; g.N = h + a->d_vd; g.K1 = h; g.K2 = a->d_vd;
            g.A1 = a->Hv; g.lda1 = a->ldh;
            g.A2 = a->V_d; g.lda2 = a->ldvd;
            g.W = a->W_d; g.ldw = h + a->d_vd; g.bias = a->b_d;
            g.C = a->out; g.ldc = a->ldout; g.act = DMPNN_ACT_NONE;
            DMPNN_TRY(launch_linear(g, s));
        }
        return DMPNN_OK;
    }

    // ---- general route: any index arrays, undirected, edge tensors in the caller's edge order ----
    // With DMPNN_F_SPLIT16 the contractions run on the f16 pipe with the exact operand split (dmpnn_rows16.hip) where
    // the shapes / alignments allow it; the weights are pre-split into `wsplit` first (or found there: WSPLIT_READY).
    const bool use16 = (a->flags & DMPNN_F_SPLIT16) != 0;
    SplitWView w16[4];
    if (use16) {
        DMPNN_CHECK_ARG(a->wsplit && a->wsplit_bytes >= steps16_wsplit_bytes(*a), "forward(split16): wsplit workspace missing or too small");
        unsigned char* wp = static_cast<unsigned char*>(a->wsplit);
        const bool ready = (a->flags & DMPNN_F_WSPLIT_READY) != 0;
        const float* Ws[4] = {a->W_i, a->W_h, a->W_o, has_vd ? a->W_d : nullptr};
        const int64_t Ns[4] = {h, h, h, h + a->d_vd}, Ks[4] = {dv + de, h, dv + h, h + a->d_vd};
        SplitWJob jobs[4];
        SplitWView views[4];
        int idx[4], nj = 0;
        for (int i = 0; i < 4; ++i) {
            if (!Ws[i]) continue;
            if (ready) w16[i] = split_weights_view_of(wp, Ns[i], Ks[i]);
            else { jobs[nj] = SplitWJob{Ws[i], Ks[i], Ns[i], Ks[i], 0, wp}; idx[nj++] = i; }
            wp += linear16_wsplit_bytes(Ns[i], Ks[i]);
        }
        DMPNN_TRY(split_weights_views(jobs, nj, views, s));   // (one launch for all of them)
        for (int k = 0; k < nj; ++k) w16[idx[k]] = views[k];
    }
    auto lin = [&](const dmpnn_gemm_args& g, int slot) -> int {
        if (use16 && linear16_ok(g)) return launch_linear16_view(g, w16[slot], nullptr, 0, s);
        return launch_linear(g, s);
    };
    // K1  H0 = W_i([V[src] || E]) (+ b_i)                         mixins.py:8-9
    {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nE; g.N = h; g.K1 = dv; g.K2 = de;
        g.A1 = a->V; g.lda1 = a->ldv; g.gather1 = pv.src; g.gather1_rows = nV;
        g.A2 = a->E; g.lda2 = a->lde;
        g.W = a->W_i; g.ldw = dv + de; g.bias = a->b_i;
        g.C = a->H0; g.ldc = a->ldh; g.act = DMPNN_ACT_NONE;
        DMPNN_TRY(lin(g, 0));
    }
    const float* Hprev = a->H0;   // H^(0) = tau(H0) is formed on load (base.py:200)
    int act_on_load = a->act;
    for (int t = 1; t < a->depth; ++t) {
        float* Mt = a->Ms ? a->Ms + ((t - 1) % a->n_mslots) * slot : nullptr;
        float* Ht = a->Hs ? a->Hs + ((t - 1) % a->n_hslots) * slot : nullptr;
        // K2  M = segsum_dst(H)[src] - H[rev]                      mixins.py:11-18 (+ base.py:202-203)
        DMPNN_TRY(launch_message(pv, nV, nE, h, Hprev, a->ldh, Mt, a->ldh, act_on_load, a->act_slope,
                                 a->act_slope_ptr, a->flags, s));
        // K3  H = tau(H0 + W_h(M))                                 base.py:135-141
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nE; g.N = h; g.K1 = h; g.K2 = 0;
        g.A1 = Mt; g.lda1 = a->ldh;
        g.W = a->W_h; g.ldw = h; g.bias = a->b_h;
        g.Cadd = a->H0; g.ldcadd = a->ldh;
        g.C = Ht; g.ldc = a->ldh;
        g.act = a->act; g.act_slope = a->act_slope; g.act_slope_ptr = a->act_slope_ptr;
        DMPNN_TRY(lin(g, 1));
        Hprev = Ht;
        act_on_load = DMPNN_ACT_NONE;
    }
    // K4  Mv = segsum_dst(H)                                       base.py:208-211
    DMPNN_TRY(launch_aggregate(pv, nV, nE, h, Hprev, a->ldh, a->Mv, a->ldh, act_on_load, a->act_slope,
                               a->act_slope_ptr, s));
    // K5  finalize                                                 base.py:180-194
    {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nV; g.N = h; g.K1 = dv; g.K2 = h;
        g.A1 = a->V; g.lda1 = a->ldv;
        g.A2 = a->Mv; g.lda2 = a->ldh;
        g.W = a->W_o; g.ldw = dv + h; g.bias = a->b_o;
        g.C = has_vd ? a->Hv : a->out; g.ldc = has_vd ? a->ldh : a->ldout;
        g.act = a->act; g.act_slope = a->act_slope; g.act_slope_ptr = a->act_slope_ptr;
        DMPNN_TRY(lin(g, 2));
    }
    if (has_vd) {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nV; g.N = h + a->d_vd; g.K1 = h; g.K2 = a->d_vd;
        g.A1 = a->Hv; g.lda1 = a->ldh;
        g.A2 = a->V_d; g.lda2 = a->ldvd;
        g.W = a->W_d; g.ldw = h + a->d_vd; g.bias = a->b_d;
        g.C = a->out; g.ldc = a->ldout; g.act = DMPNN_ACT_NONE;  // no tau on the W_d branch (base.py:187-188)
        DMPNN_TRY(lin(g, 3));
    }
    return DMPNN_OK;
}

}  // extern "C"
