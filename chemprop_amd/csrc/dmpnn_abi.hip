// extern "C" surface of libdmpnn (see include/dmpnn.h) and the forward driver that chains the
// row kernels exactly as chemprop/nn/message_passing/base.py:196-212 chains its ATen ops.
#include <stdarg.h>
#include <string.h>

#include "dmpnn_common.hpp"

namespace dmpnn {

static thread_local char g_err[512] = "";
static thread_local int g_launches = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void count_launch() { ++g_launches; }

static int check_graph_sizes(int64_t nV, int64_t nE) {
    DMPNN_CHECK_ARG(nV >= 0 && nE >= 0, "negative graph size (n_atoms=%lld n_edges=%lld)", (long long)nV, (long long)nE);
    DMPNN_CHECK_ARG(nV < (int64_t(1) << 31) - 64 && nE < (int64_t(1) << 31) - 64,
                    "graph too large for int32 indices (n_atoms=%lld n_edges=%lld)", (long long)nV, (long long)nE);
    return DMPNN_OK;
}

}  // namespace dmpnn

using namespace dmpnn;

extern "C" {

int dmpnn_version(void) { return DMPNN_ABI_VERSION; }
const char* dmpnn_last_error_string(void) { return g_err; }
int dmpnn_last_launch_count(void) { return g_launches; }

size_t dmpnn_plan_bytes(int64_t n_atoms, int64_t n_edges) {
    if (n_atoms < 0 || n_edges < 0) return 0;
    return (size_t)plan_layout(n_atoms, n_edges).words * sizeof(int);
}

int dmpnn_plan_layout(int64_t n_atoms, int64_t n_edges, int64_t off[5]) {
    DMPNN_CHECK_ARG(off != nullptr, "plan_layout: null output");
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    const PlanLayout L = plan_layout(n_atoms, n_edges);
    off[0] = L.src; off[1] = L.dst; off[2] = L.rev; off[3] = L.row_ptr; off[4] = L.perm;
    return DMPNN_OK;
}

int dmpnn_prepare(const int64_t* edge_index, const int64_t* rev, int64_t n_atoms, int64_t n_edges,
                  void* plan, size_t plan_bytes, void* stream) {
    DMPNN_TRY(check_graph_sizes(n_atoms, n_edges));
    DMPNN_CHECK_ARG(plan != nullptr, "prepare: null plan");
    DMPNN_CHECK_ARG(n_edges == 0 || (edge_index && rev), "prepare: null index arrays");
    if (plan_bytes < dmpnn_plan_bytes(n_atoms, n_edges)) {
        set_error("prepare: plan buffer too small (%zu < %zu bytes)", plan_bytes, dmpnn_plan_bytes(n_atoms, n_edges));
        return DMPNN_ENOSPC;
    }
    DMPNN_CHECK_ARG(aligned16(plan), "prepare: plan must be 16-byte aligned");
    return launch_prepare(edge_index, rev, n_atoms, n_edges, static_cast<int*>(plan), static_cast<hipStream_t>(stream));
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
    DMPNN_CHECK_ARG(nE == 0 || (a->E && a->H0), "forward: null E / H0");
    const bool has_vd = a->W_d != nullptr;
    DMPNN_CHECK_ARG(!has_vd || (a->d_vd > 0 && a->V_d && a->b_d && a->Hv && a->ldvd >= a->d_vd),
                    "forward: W_d given but V_d / b_d / Hv / d_vd missing");
    DMPNN_CHECK_ARG(a->ldout >= h + (has_vd ? a->d_vd : 0), "forward: ldout too small");
    if (a->depth > 1 && nE > 0)
        DMPNN_CHECK_ARG(a->Hs && a->Ms && a->n_hslots >= 1 && a->n_mslots >= 1, "forward: missing Hs / Ms workspace");

    hipStream_t s = static_cast<hipStream_t>(stream);
    const PlanView pv = plan_view(a->plan, nV, nE);
    const int64_t slot = nE * a->ldh;

    // K1  H0 = W_i([V[src] || E]) (+ b_i)                         mixins.py:8-9
    {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nE; g.N = h; g.K1 = dv; g.K2 = de;
        g.A1 = a->V; g.lda1 = a->ldv; g.gather1 = pv.src;
        g.A2 = a->E; g.lda2 = a->lde;
        g.W = a->W_i; g.ldw = dv + de; g.bias = a->b_i;
        g.C = a->H0; g.ldc = a->ldh; g.act = DMPNN_ACT_NONE;
        DMPNN_TRY(launch_linear(g, s));
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
        DMPNN_TRY(launch_linear(g, s));
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
        DMPNN_TRY(launch_linear(g, s));
    }
    if (has_vd) {
        dmpnn_gemm_args g;
        memset(&g, 0, sizeof(g));
        g.M = nV; g.N = h + a->d_vd; g.K1 = h; g.K2 = a->d_vd;
        g.A1 = a->Hv; g.lda1 = a->ldh;
        g.A2 = a->V_d; g.lda2 = a->ldvd;
        g.W = a->W_d; g.ldw = h + a->d_vd; g.bias = a->b_d;
        g.C = a->out; g.ldc = a->ldout; g.act = DMPNN_ACT_NONE;  // no tau on the W_d branch (base.py:187-188)
        DMPNN_TRY(launch_linear(g, s));
    }
    return DMPNN_OK;
}

}  // extern "C"
