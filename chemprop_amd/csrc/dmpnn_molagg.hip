// f1 — per-molecule aggregation of the atom representations (the step after the path inside
// MPNN.fingerprint, models/model.py:131).
//
// Reference (chemprop/nn/agg.py:66-113): Mean / Sum / Norm aggregation build an [V, h] int64 index
// (`batch.unsqueeze(1).repeat(1, h)`), read `batch.max()` on the host and call
//     zeros(n_mols, h).scatter_reduce_(0, index, H, "sum" | "mean", include_self=False)    (/ norm)
// i.e. a segment sum over a SORTED index (`batch` is non-decreasing: data/collate.py:48-62).
//
// MI355X form (HBM bound: every H row read once, every output row written once):
//   k_mol_bounds  one pass over `batch`: first / one-past-last atom of every molecule, and the
//                 invariants (ids in range, non-decreasing) decided on device -> flag word;
//   k_mol_reduce  one wave per (molecule, 256-column slab): rows added in increasing atom order
//                 (the reference's sequential scatter order: bit-exact), then `/ count` or `/ norm`
//                 with a true division like the reference; molecules without atoms give zero rows
//                 (agg.py:44-46); an invalid `batch` poisons the output with NaN;
//   k_mol_bwd     gH[v] = gOut[batch[v]] (/ count | / norm): the transpose, one wave per atom row.
#include "dmpnn_common.hpp"

namespace dmpnn {
namespace {

enum : int { MOLAGG_RANGE = 1, MOLAGG_UNSORTED = 2 };

// ws: int first[n_mols] | int end[n_mols] | int flag   (zero-initialised by the caller of k_mol_bounds)
__global__ void k_mol_bounds(const int64_t* __restrict__ batch, int64_t nV, int64_t n_mols, int* __restrict__ ws) {
    int* first = ws;
    int* end = ws + n_mols;
    int* flag = ws + 2 * n_mols;
    int bad = 0;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nV; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = batch[v];
        const int64_t prev = v > 0 ? batch[v - 1] : -1;
        const int64_t next = v + 1 < nV ? batch[v + 1] : n_mols;
        if (b < 0 || b >= n_mols) { bad |= MOLAGG_RANGE; continue; }
        if (b < prev) bad |= MOLAGG_UNSORTED;
        if (b != prev) first[b] = (int)v;
        if (b != next) end[b] = (int)(v + 1);
    }
    if (bad) atomicOr(flag, bad);
}

// The same pass by ONE workgroup that zeroes its own tables first (no memset launch in front of it): a batch of up to 32 768 atoms.
constexpr int64_t kBoundsOneMaxAtoms = 32768;
__global__ __launch_bounds__(1024) void k_mol_bounds_one(const int64_t* __restrict__ batch, int64_t nV, int64_t n_mols, int* __restrict__ ws) {
    for (int64_t i = threadIdx.x; i < 3 * n_mols + 4; i += 1024) ws[i] = 0;
    __threadfence();   // (the zeroes are in memory before any thread of this workgroup writes a bound over them)
    __syncthreads();
    int* first = ws;
    int* end = ws + n_mols;
    int bad = 0;
    for (int64_t v = threadIdx.x; v < nV; v += 1024) {
        const int64_t b = batch[v];
        const int64_t prev = v > 0 ? batch[v - 1] : -1;
        const int64_t next = v + 1 < nV ? batch[v + 1] : n_mols;
        if (b < 0 || b >= n_mols) { bad |= MOLAGG_RANGE; continue; }
        if (b < prev) bad |= MOLAGG_UNSORTED;
        if (b != prev) first[b] = (int)v;
        if (b != next) end[b] = (int)(v + 1);
    }
    if (bad) atomicOr(ws + 2 * n_mols, bad);
}

struct MolAggArgs {
    const float* H; int64_t ldh;
    float* out; int64_t ldo;
    const int* ws;
    int64_t n_mols;
    int h, mode;
    float norm;
};

template <int VEC>
__global__ __launch_bounds__(256) void k_mol_reduce(MolAggArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int slabs = (a.h + 64 * VEC - 1) / (64 * VEC);
    const int64_t m = wave / slabs;
    if (m >= a.n_mols) return;
    const int c = (int)(wave - m * slabs) * 64 * VEC + lane * VEC;
    const int flag = a.ws[2 * a.n_mols];
    const int v0 = a.ws[m], v1 = a.ws[a.n_mols + m];
    if (c >= a.h) return;
    float acc[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[t] = 0.f;
    for (int v = v0; v < v1; ++v) {
        const float* p = a.H + (int64_t)v * a.ldh + c;
        if (VEC == 4) {
            const float4 x = *reinterpret_cast<const float4*>(p);
            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = v == v0 ? xs[t] : acc[t] + xs[t];  // include_self=False: the first addend is copied
        } else {
            acc[0] = v == v0 ? p[0] : acc[0] + p[0];
        }
    }
    const float cnt = (float)(v1 - v0);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
        float y = acc[t];
        if (a.mode == DMPNN_MOLAGG_MEAN && v1 > v0) y = y / cnt;
        if (a.mode == DMPNN_MOLAGG_NORM) y = y / a.norm;
        if (flag) y = __int_as_float(0x7fc00000);
        if (c + t < a.h) a.out[m * a.ldo + c + t] = y;
    }
}

struct MolBwdArgs {
    const float* gout; int64_t ldg;
    const int64_t* batch;
    float* gH; int64_t ldgh;
    const int* ws;
    int64_t nV, n_mols;
    int h, mode;
    float norm;
};

__global__ __launch_bounds__(256) void k_mol_bwd(MolBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t v = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (v >= a.nV) return;
    const int flag = a.ws[2 * a.n_mols];
    int64_t m = a.batch[v];
    const bool ok = m >= 0 && m < a.n_mols;
    m = ok ? m : 0;
    const float cnt = (float)(a.ws[a.n_mols + m] - a.ws[m]);
    for (int c = lane; c < a.h; c += 64) {
        float g = a.gout[m * a.ldg + c];
        if (a.mode == DMPNN_MOLAGG_MEAN) g = g / cnt;
        if (a.mode == DMPNN_MOLAGG_NORM) g = g / a.norm;
        if (flag || !ok) g = __int_as_float(0x7fc00000);
        a.gH[v * a.ldgh + c] = g;
    }
}

}  // namespace
}  // namespace dmpnn

using namespace dmpnn;

extern "C" {

// first[n] | end[n] | flag, 3 words of padding | done[n] (round 5: 1 where the forward tile kernel already wrote the molecule's aggregate;
// zeroed with the rest of the table — the head's column kernel reads it, dmpnn_head.hip)
size_t dmpnn_molagg_ws_bytes(int64_t n_mols) { return (size_t)(3 * (n_mols > 0 ? n_mols : 0) + 4) * sizeof(int); }

int dmpnn_molagg_bounds(const int64_t* batch, int64_t n_atoms, int64_t n_mols, void* ws, size_t ws_bytes, void* stream) {
    DMPNN_CHECK_ARG(n_atoms >= 0 && n_mols >= 0 && n_atoms < (1ll << 31) && n_mols < (1ll << 30), "molagg_bounds: sizes out of range");
    DMPNN_CHECK_ARG(ws && ws_bytes >= dmpnn_molagg_ws_bytes(n_mols), "molagg_bounds: workspace missing or too small");
    DMPNN_CHECK_ARG(n_atoms == 0 || batch, "molagg_bounds: batch is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_atoms > 0 && n_atoms <= kBoundsOneMaxAtoms) {
        hipLaunchKernelGGL(k_mol_bounds_one, dim3(1), dim3(1024), 0, s, batch, n_atoms, n_mols, static_cast<int*>(ws));
        DMPNN_CHECK_LAUNCH("k_mol_bounds_one");
        return DMPNN_OK;
    }
    if (hipMemsetAsync(ws, 0, dmpnn_molagg_ws_bytes(n_mols), s) != hipSuccess) {
        set_error("molagg_bounds: hipMemsetAsync failed");
        return DMPNN_EHIP;
    }
    if (n_atoms == 0) return DMPNN_OK;
    int64_t blocks = (n_atoms + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_mol_bounds, dim3((unsigned)blocks), dim3(256), 0, s, batch, n_atoms, n_mols, static_cast<int*>(ws));
    DMPNN_CHECK_LAUNCH("k_mol_bounds");
    return DMPNN_OK;
}

int dmpnn_molagg_fwd(const float* H, int64_t ldh, int64_t n_atoms, int64_t d_h, int64_t n_mols, const void* ws, int mode,
                     float norm, float* out, int64_t ldo, void* stream) {
    DMPNN_CHECK_ARG(mode == DMPNN_MOLAGG_SUM || mode == DMPNN_MOLAGG_MEAN || mode == DMPNN_MOLAGG_NORM, "molagg_fwd: unknown mode %d", mode);
    DMPNN_CHECK_ARG(d_h >= 0 && d_h < (1 << 24) && ldh >= d_h && ldo >= d_h, "molagg_fwd: bad row sizes");
    if (n_mols == 0 || d_h == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(ws && out && (n_atoms == 0 || H), "molagg_fwd: NULL pointer");
    MolAggArgs a{H, ldh, out, ldo, static_cast<const int*>(ws), n_mols, (int)d_h, mode, norm};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = d_h % 4 == 0 && ldh % 4 == 0 && aligned16(H);
    const int per_wave = vec ? 256 : 64;
    const int64_t waves = n_mols * ((d_h + per_wave - 1) / per_wave);
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (vec) hipLaunchKernelGGL(k_mol_reduce<4>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_mol_reduce<1>, grid, dim3(256), 0, s, a);
    DMPNN_CHECK_LAUNCH("k_mol_reduce");
    return DMPNN_OK;
}

int dmpnn_molagg_bwd(const float* gout, int64_t ldg, const int64_t* batch, int64_t n_atoms, int64_t d_h, int64_t n_mols,
                     const void* ws, int mode, float norm, float* gH, int64_t ldgh, void* stream) {
    DMPNN_CHECK_ARG(mode == DMPNN_MOLAGG_SUM || mode == DMPNN_MOLAGG_MEAN || mode == DMPNN_MOLAGG_NORM, "molagg_bwd: unknown mode %d", mode);
    DMPNN_CHECK_ARG(d_h >= 0 && ldg >= d_h && ldgh >= d_h, "molagg_bwd: bad row sizes");
    if (n_atoms == 0 || d_h == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(n_mols > 0 && ws && gout && batch && gH, "molagg_bwd: NULL pointer / no molecules");
    MolBwdArgs a{gout, ldg, batch, gH, ldgh, static_cast<const int*>(ws), n_atoms, n_mols, (int)d_h, mode, norm};
    hipLaunchKernelGGL(k_mol_bwd, dim3((unsigned)((n_atoms + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    DMPNN_CHECK_LAUNCH("k_mol_bwd");
    return DMPNN_OK;
}

}  // extern "C"

// ---- generic row gather: out[i] = X[idx[i]] (f2, atom messages: M[e] = S[src(e)], mixins.py:30) ----------
namespace dmpnn {
namespace {
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ X, int64_t ldx, int64_t n_src,
                                                     const int* __restrict__ idx, int64_t n_out, int d,
                                                     float* __restrict__ out, int64_t ldo, int vec) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n_out; i += gridDim.x * 4ll) {
        const int64_t r = idx[i];
        const bool ok = r >= 0 && r < n_src;  // an index out of range gives a NaN row, not a wild read
        const float* src = X + (ok ? r : 0) * ldx;
        float* dst = out + i * ldo;
        if (vec) {
            for (int c = lane * 4; c < d; c += 256) {
                float4 v = *reinterpret_cast<const float4*>(src + c);
                if (!ok) v.x = v.y = v.z = v.w = __int_as_float(0x7fc00000);
                *reinterpret_cast<float4*>(dst + c) = v;
            }
        } else {
            for (int c = lane; c < d; c += 64) dst[c] = ok ? src[c] : __int_as_float(0x7fc00000);
        }
    }
}
}  // namespace
}  // namespace dmpnn

extern "C" int dmpnn_gather_rows(const float* X, int64_t ldx, int64_t n_src, const int* idx, int64_t n_out, int64_t d,
                                 float* out, int64_t ldo, void* stream) {
    DMPNN_CHECK_ARG(d >= 0 && d < (1 << 24) && ldx >= d && ldo >= d && n_src >= 0 && n_out >= 0, "gather_rows: bad sizes");
    if (n_out == 0 || d == 0) return DMPNN_OK;
    DMPNN_CHECK_ARG(X && idx && out, "gather_rows: NULL pointer");
    const int vec = d % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && dmpnn::aligned16(X) && dmpnn::aligned16(out);
    int64_t blocks = (n_out + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dmpnn::k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, n_src,
                       idx, n_out, (int)d, out, ldo, vec);
    DMPNN_CHECK_LAUNCH("k_gather_rows");
    return DMPNN_OK;
}
