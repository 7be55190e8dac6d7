// Host side of the per-step split-MFMA contraction (dmpnn_rows16_impl.hpp): weight pre-split, shape / alignment
// gate, launch; C entry points dmpnn_linear16_*.
#include <string.h>

#include "dmpnn_rows16_impl.hpp"

namespace dmpnn {
namespace rows16 {
DMPNN_DEFINE_ROWS16(1, 4)
DMPNN_DEFINE_ROWS16(2, 4)
DMPNN_DEFINE_ROWS16(3, 4)
DMPNN_DEFINE_ROWS16(4, 4)
DMPNN_DEFINE_ROWS16(5, 4)
DMPNN_DEFINE_ROWS16(1, 12)
DMPNN_DEFINE_ROWS16(2, 12)
DMPNN_DEFINE_ROWS16(3, 12)
DMPNN_DEFINE_ROWS16(4, 12)
DMPNN_DEFINE_ROWS16(5, 12)
}  // namespace rows16

static size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

// bytes of one pre-split matrix [N, K] (fragment-major) + its inverse column scales
size_t linear16_wsplit_bytes(int64_t N, int64_t K) {
    const size_t NT = (size_t)(N + 15) / 16, nc = (size_t)(K + 31) / 32;
    return al256(NT * nc * 2048) + al256((size_t)N * 4);
}

// pre-split W [N, K] (tr: given as [K, N]) into ws; returns the descriptor the kernels read
int split_weights(const float* W, int64_t ldw, int64_t N, int64_t K, int tr, void* ws, mega16::SplitW* out, hipStream_t s) {
    const size_t NT = (size_t)(N + 15) / 16, nc = (size_t)(K + 31) / 32;
    unsigned char* p = static_cast<unsigned char*>(ws);
    float* inv = reinterpret_cast<float*>(p + al256(NT * nc * 2048));
    mega16::SplitArgs sp;
    memset(&sp, 0, sizeof(sp));
    sp.N = (int)N; sp.n_jobs = 1;
    sp.job[0] = mega16::SplitJob{W, (int)ldw, 0, (int)K, 0, (int)K, p, (int)nc, inv, tr};
    const unsigned waves = (unsigned)(((N + 15) / 16) * 16);
    hipLaunchKernelGGL(mega16::k_split_weights, dim3((waves + 3) / 4), dim3(256), 0, s, sp);
    DMPNN_CHECK_LAUNCH("k_split_weights");
    out->p = p; out->inv_scale = inv; out->nc = (int)nc;
    return DMPNN_OK;
}

// shapes / alignments the split kernel takes (else the caller stays on the fp32-MFMA kernel)
bool linear16_ok(const dmpnn_gemm_args& a) {
    if (a.M <= 0 || a.N <= 0 || a.K1 + a.K2 <= 0) return false;
    if (a.K1 % 2 || a.K2 % 2) return false;  // operands are read as pairs of floats
    auto ok8 = [](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 7u) == 0 && ld % 2 == 0); };
    if (!ok8(a.A1, a.lda1) || !ok8(a.A2, a.lda2)) return false;
    if (a.K1 == 0) return false;  // (single operand in the A2 slot: the caller swaps it into A1)
    if (a.M >= (int64_t(1) << 31) / 64 * 48) return false;
    if (a.gather1 && a.gather1_rows * a.lda1 * 4 > 0x7FFFFFFF) return false;
    return true;
}

int launch_linear16(const dmpnn_gemm_args& a, const mega16::SplitW& W, const int* poison_flags, int poison_mask, hipStream_t s) {
    if (a.M == 0 || a.N == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(linear16_ok(a), "linear16: shapes / alignment not supported by the split kernel");
    rows16::Rows16K g;
    memset(&g, 0, sizeof(g));
    g.M = (int)a.M; g.N = (int)a.N; g.K1 = (int)a.K1; g.K2 = (int)a.K2;
    g.A1 = a.A1; g.lda1 = (int)a.lda1; g.gather1 = a.gather1;
    g.a1_bytes = a.gather1 ? (unsigned)(a.gather1_rows * a.lda1 * 4) : 0u;
    g.A2 = a.K2 ? a.A2 : nullptr; g.lda2 = (int)a.lda2;
    g.W = W; g.bias = a.bias; g.Cadd = a.Cadd; g.ldcadd = (int)a.ldcadd;
    g.C = a.C; g.ldc = (int)a.ldc; g.Zpre = a.Zpre; g.ldz = (int)a.ldz;
    g.act = a.act; g.slope = a.act_slope; g.slope_ptr = a.act_slope_ptr;
    g.poison_flags = poison_flags; g.poison_mask = poison_mask;
    auto v16 = [](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15u) == 0 && ld % 4 == 0); };
    g.vec_out = (a.N % 4 == 0 && v16(a.C, a.ldc) && v16(a.Zpre, a.ldz) && v16(a.Cadd, a.ldcadd)) ? 1 : 0;
    // column blocks of at most 320 columns (5 MFMA column tiles per wave: the register budget of two waves per SIMD),
    // balanced: d_h = 512 -> 2 x 256
    const int col_blocks = (int)((a.N + 319) / 320);
    const int WN = (int)((a.N + 64 * col_blocks - 1) / (64 * col_blocks));
    const int row_tiles = (int)((a.M + rows16::BM - 1) / rows16::BM);
    // about one tile per CU: the whole (<= 384-column) operand row in one group, one workgroup per CU
    const bool one_group = (int64_t)row_tiles * col_blocks <= 512;
    if (one_group) {
        switch (WN) {
            case 1: return rows16::launch_rows16<1, 12>(g, row_tiles, col_blocks, s);
            case 2: return rows16::launch_rows16<2, 12>(g, row_tiles, col_blocks, s);
            case 3: return rows16::launch_rows16<3, 12>(g, row_tiles, col_blocks, s);
            case 4: return rows16::launch_rows16<4, 12>(g, row_tiles, col_blocks, s);
            default: return rows16::launch_rows16<5, 12>(g, row_tiles, col_blocks, s);
        }
    }
    switch (WN) {
        case 1: return rows16::launch_rows16<1, 4>(g, row_tiles, col_blocks, s);
        case 2: return rows16::launch_rows16<2, 4>(g, row_tiles, col_blocks, s);
        case 3: return rows16::launch_rows16<3, 4>(g, row_tiles, col_blocks, s);
        case 4: return rows16::launch_rows16<4, 4>(g, row_tiles, col_blocks, s);
        default: return rows16::launch_rows16<5, 4>(g, row_tiles, col_blocks, s);
    }
}

// several matrices in ONE launch (<= 6: the per-step routes' W_i | W_h | W_o | W_d | W_o[:, d_v:] | W_o[:, :d_v] — each used to be a
// launch of its own, hidden behind the version-keyed cache of rounds 1-3; without the cache they are one launch per forward)
// ... the argument block only (views filled in): for a caller whose first kernel does not read the weights and lets the split ride in
// that launch (launch_fused16_forward: k_split_rows_w)
bool split_weights_args(const SplitWJob* jobs, int n, SplitWView* out, mega16::SplitArgs* spp) {
    if (n <= 0 || n > 6) return false;
    mega16::SplitArgs& sp = *spp;
    memset(&sp, 0, sizeof(sp));
    int64_t Nmax = 0;
    for (int i = 0; i < n; ++i) {
        const SplitWJob& j = jobs[i];
        const size_t NT = (size_t)(j.N + 15) / 16, nc = (size_t)(j.K + 31) / 32;
        unsigned char* p = static_cast<unsigned char*>(j.ws);
        float* inv = reinterpret_cast<float*>(p + al256(NT * nc * 2048));
        sp.job[i] = mega16::SplitJob{j.W, (int)j.ldw, 0, (int)j.K, 0, (int)j.K, p, (int)nc, inv, j.tr, (int)j.N};
        out[i].p = p; out[i].inv_scale = inv; out[i].nc = (int)nc;
        if (j.N > Nmax) Nmax = j.N;
    }
    sp.N = (int)Nmax; sp.n_jobs = n;
    return true;
}
int launch_split_args(const mega16::SplitArgs& sp, hipStream_t s) {
    const unsigned waves = (unsigned)(((sp.N + 15) / 16) * 16) * (unsigned)sp.n_jobs;
    hipLaunchKernelGGL(mega16::k_split_weights, dim3((waves + 3) / 4), dim3(256), 0, s, sp);
    DMPNN_CHECK_LAUNCH("k_split_weights");
    return DMPNN_OK;
}
int split_weights_views(const SplitWJob* jobs, int n, SplitWView* out, hipStream_t s) {
    if (n <= 0) return DMPNN_OK;
    if (n > 6) { set_error("split_weights_views: at most 6 matrices per launch"); return DMPNN_EINVAL; }
    mega16::SplitArgs sp;
    memset(&sp, 0, sizeof(sp));
    int64_t Nmax = 0;
    for (int i = 0; i < n; ++i) {
        const SplitWJob& j = jobs[i];
        const size_t NT = (size_t)(j.N + 15) / 16, nc = (size_t)(j.K + 31) / 32;
        unsigned char* p = static_cast<unsigned char*>(j.ws);
        float* inv = reinterpret_cast<float*>(p + al256(NT * nc * 2048));
        sp.job[i] = mega16::SplitJob{j.W, (int)j.ldw, 0, (int)j.K, 0, (int)j.K, p, (int)nc, inv, j.tr, (int)j.N};
        out[i].p = p; out[i].inv_scale = inv; out[i].nc = (int)nc;
        if (j.N > Nmax) Nmax = j.N;
    }
    sp.N = (int)Nmax; sp.n_jobs = n;
    const unsigned waves = (unsigned)(((Nmax + 15) / 16) * 16) * (unsigned)n;
    hipLaunchKernelGGL(mega16::k_split_weights, dim3((waves + 3) / 4), dim3(256), 0, s, sp);
    DMPNN_CHECK_LAUNCH("k_split_weights");
    return DMPNN_OK;
}
int split_weights_view(const float* W, int64_t ldw, int64_t N, int64_t K, int tr, void* ws, SplitWView* out, hipStream_t s) {
    mega16::SplitW w;
    DMPNN_TRY(split_weights(W, ldw, N, K, tr, ws, &w, s));
    out->p = w.p; out->inv_scale = w.inv_scale; out->nc = w.nc;
    return DMPNN_OK;
}
SplitWView split_weights_view_of(void* ws, int64_t N, int64_t K) {
    const size_t NT = (size_t)(N + 15) / 16, nc = (size_t)(K + 31) / 32;
    SplitWView v;
    v.p = static_cast<unsigned char*>(ws);
    v.inv_scale = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + al256(NT * nc * 2048));
    v.nc = (int)nc;
    return v;
}
int launch_linear16_view(const dmpnn_gemm_args& a, const SplitWView& W, const int* poison_flags, int poison_mask, hipStream_t s) {
    mega16::SplitW w;
    w.p = W.p; w.inv_scale = W.inv_scale; w.nc = W.nc;
    return launch_linear16(a, w, poison_flags, poison_mask, s);
}

}  // namespace dmpnn

using namespace dmpnn;

extern "C" {

size_t dmpnn_linear16_wsplit_bytes(int64_t N, int64_t K) { return linear16_wsplit_bytes(N, K); }

int dmpnn_linear16_fwd(const dmpnn_gemm_args* a, void* wsplit, size_t wsplit_bytes, int wsplit_ready, void* stream) {
    DMPNN_CHECK_ARG(a && a->W, "linear16: null args");
    DMPNN_CHECK_ARG(a->M >= 0 && a->N > 0 && a->K1 >= 0 && a->K2 >= 0 && a->K1 + a->K2 > 0, "linear16: bad sizes");
    dmpnn_gemm_args g = *a;
    if (g.K1 == 0) {  // single operand lives in the A1 slot
        DMPNN_CHECK_ARG(!g.gather1, "linear16: gather without A1");
        g.A1 = g.A2; g.lda1 = g.lda2; g.K1 = g.K2; g.A2 = nullptr; g.K2 = 0; g.lda2 = 0;
    }
    const int64_t K = g.K1 + g.K2;
    DMPNN_CHECK_ARG(wsplit && wsplit_bytes >= linear16_wsplit_bytes(g.N, K), "linear16: wsplit workspace missing or too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    mega16::SplitW W;
    if (!wsplit_ready) {
        DMPNN_TRY(split_weights(g.W, g.ldw, g.N, K, 0, wsplit, &W, s));
    } else {
        const size_t NT = (size_t)(g.N + 15) / 16, nc = (size_t)(K + 31) / 32;
        W.p = static_cast<unsigned char*>(wsplit);
        W.inv_scale = reinterpret_cast<float*>(static_cast<unsigned char*>(wsplit) + al256(NT * nc * 2048));
        W.nc = (int)nc;
    }
    return launch_linear16(g, W, nullptr, 0, s);
}

int dmpnn_linear16_ok(const dmpnn_gemm_args* a) {
    if (!a) return 0;
    dmpnn_gemm_args g = *a;
    if (g.K1 == 0) { g.A1 = g.A2; g.lda1 = g.lda2; g.K1 = g.K2; g.A2 = nullptr; g.K2 = 0; g.lda2 = 0; }
    return linear16_ok(g) ? 1 : 0;
}

}  // extern "C"
