// K4 -- multiplicative update of the traces on the hoisted Gram matrices.
//
// Reference: DeformableNMF.update_temporal (Demix/dNMF.py:143-148) inside the loop of update_footprints
// (dNMF.py:172-173).  The reference recomputes G_t = A_t^T A_t and r_t = A_t^T y_t in every one of the
// iter_c rounds although neither depends on C; here they come from K3 once and only
//   c_t <- c_t * (r_t + gamma*nbr_t) / (G_t c_t + 2 gamma c_t + 1e-32)
// is iterated.  One workgroup per frame, thread k owns c[k]; arithmetic in fp64 (numpy's dtype there).
#include "common.hpp"

namespace dnmf {

// iters rounds, no neighbour term: everything stays in registers / LDS.  G is symmetric, so thread k reads
// column k (G[l][k], coalesced across threads) instead of row k.  KREG > 0: the column lives in registers.
template <int KREG>
__global__ __launch_bounds__(256) void mu_temporal_kernel(const float *__restrict__ G, const float *__restrict__ r,
                                                          float *__restrict__ C, long ldc, int K, int iters) {
    __shared__ double cs[256];
    const int t = blockIdx.x;
    const int k = threadIdx.x;
    const float *Gt = G + (long)t * K * K;
    const bool live = k < K;
    float g[KREG > 0 ? KREG : 1];
    if (KREG > 0) {
#pragma unroll
        for (int l = 0; l < KREG; ++l) g[l] = (live && l < K) ? Gt[(long)l * K + k] : 0.0f;
    }
    const double rk = live ? (double)r[(long)t * K + k] : 0.0;
    double c = live ? (double)C[(long)k * ldc + t] : 0.0;
    for (int it = 0; it < iters; ++it) {
        cs[k] = c;
        __syncthreads();
        double dot = 0.0;
        if (KREG > 0) {
#pragma unroll
            for (int l = 0; l < KREG; ++l) dot = fma((double)g[l], cs[l], dot);
        } else {
            for (int l = 0; l < K; ++l) dot = fma(live ? (double)Gt[(long)l * K + k] : 0.0, cs[l], dot);
        }
        c = (c * rk) / (dot + 1e-32);
        __syncthreads();
    }
    if (live) C[(long)k * ldc + t] = (float)c;
}

// The same rounds when G has a known pattern (K3n: footprints whose boxes never meet have G = 0 for every warp).
// nbr (K,NN): for row k the columns that can be non-zero, ascending, padded with columns outside the pattern.  The
// dense kernel adds the terms in ascending column order and a term with G = 0 adds an exact zero, so the result is
// bit-identical to mu_temporal_kernel's; only the work shrinks from K to NN terms per row and round.
template <int NN>
__global__ __launch_bounds__(256) void mu_temporal_nbr_kernel(const float *__restrict__ G, const float *__restrict__ r,
                                                              float *__restrict__ C, long ldc, int K, int iters,
                                                              const int *__restrict__ nbr) {
    __shared__ double cs[256];
    const int t = blockIdx.x;
    const int k = threadIdx.x;
    const float *Gt = G + (long)t * K * K;
    const bool live = k < K;
    double g[NN];
    int li[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        li[j] = live ? nbr[k * NN + j] : 0;
        g[j] = live ? (double)Gt[(long)li[j] * K + k] : 0.0;
    }
    const double rk = live ? (double)r[(long)t * K + k] : 0.0;
    double c = live ? (double)C[(long)k * ldc + t] : 0.0;
    for (int it = 0; it < iters; ++it) {
        cs[k] = c;
        __syncthreads();
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < NN; ++j) dot = fma(g[j], cs[li[j]], dot);
        c = (c * rk) / (dot + 1e-32);
        __syncthreads();
    }
    if (live) C[(long)k * ldc + t] = (float)c;
}

// The same again, reading the slot tables K3n left in its workspace instead of a dense G: entry (k,l) is the ordered
// sum over the chunks of slot pair_slot[k][l] -- exactly what gram_lists_finish_kernel would have written -- and r[k]
// the sum of slot k.  Saves writing and re-reading (T,K,K).
template <int NN>
__global__ __launch_bounds__(256) void mu_temporal_slots_kernel(const float *__restrict__ slab, int nchunks, int nslot,
                                                                const int *__restrict__ pair_slot,
                                                                float *__restrict__ C, long ldc, int K, int iters,
                                                                const int *__restrict__ nbr) {
    __shared__ double cs[256];
    const int t = blockIdx.x;
    const int k = threadIdx.x;
    const float *src = slab + (long)t * nchunks * nslot;
    const bool live = k < K;
    auto slot_sum = [&](int slot) {
        float s = 0.0f;
        if (slot != nslot - 1)
            for (int c = 0; c < nchunks; ++c) s += src[(long)c * nslot + slot];
        return s;
    };
    double g[NN];
    int li[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        li[j] = live ? nbr[k * NN + j] : 0;
        g[j] = live ? (double)slot_sum(pair_slot[(long)li[j] * K + k]) : 0.0;
    }
    const double rk = live ? (double)slot_sum(k) : 0.0;
    double c = live ? (double)C[(long)k * ldc + t] : 0.0;
    for (int it = 0; it < iters; ++it) {
        cs[k] = c;
        __syncthreads();
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < NN; ++j) dot = fma(g[j], cs[li[j]], dot);
        c = (c * rk) / (dot + 1e-32);
        __syncthreads();
    }
    if (live) C[(long)k * ldc + t] = (float)c;
}

// one round with the temporal-smoothness term on an fp64 state (Jacobi: all of Cin is the old iterate)
__global__ __launch_bounds__(256) void mu_temporal_step_kernel(const float *__restrict__ G, const float *__restrict__ r,
                                                               const double *__restrict__ Cin, double *__restrict__ Cout,
                                                               long ldc, int K, int T, double gamma,
                                                               const double *__restrict__ c_left,
                                                               const double *__restrict__ c_right) {
    __shared__ double cs[256];
    const int t = blockIdx.x;
    const int k = threadIdx.x;
    const float *Gt = G + (long)t * K * K;
    const bool live = k < K;
    const double c = live ? Cin[(long)k * ldc + t] : 0.0;
    cs[k] = c;
    __syncthreads();
    if (!live) return;
    double dot = 0.0;
    for (int l = 0; l < K; ++l) dot = fma((double)Gt[(long)l * K + k], cs[l], dot);
    const double left = t > 0 ? Cin[(long)k * ldc + t - 1] : (c_left ? c_left[k] : c);
    const double right = t + 1 < T ? Cin[(long)k * ldc + t + 1] : (c_right ? c_right[k] : c);
    const double c1 = (double)r[(long)t * K + k] + gamma * (left + right);
    const double c2 = dot + 2.0 * gamma * c;
    Cout[(long)k * ldc + t] = (c * c1) / (c2 + 1e-32);
}

}  // namespace dnmf

extern "C" {

int dnmf_mu_temporal(const float *G, const float *r, float *C, long ldc, int K, int T, int iters,
                     dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(G && r && C, DNMF_E_NULL, "dnmf_mu_temporal: NULL buffer");
    DNMF_REQUIRE(K > 0 && T > 0 && ldc >= T && iters >= 0, DNMF_E_SHAPE, "dnmf_mu_temporal: K=%d T=%d ldc=%ld iters=%d", K,
                 T, ldc, iters);
    DNMF_REQUIRE(K <= 256, DNMF_E_UNSUPPORTED, "dnmf_mu_temporal: K=%d > 256 (not built yet)", K);
    if (iters == 0) return DNMF_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)T);
    if (K <= 32)
        hipLaunchKernelGGL(mu_temporal_kernel<32>, grid, dim3(64), 0, st, G, r, C, ldc, K, iters);
    else if (K <= 64)
        hipLaunchKernelGGL(mu_temporal_kernel<64>, grid, dim3(64), 0, st, G, r, C, ldc, K, iters);
    else if (K <= 128)
        hipLaunchKernelGGL(mu_temporal_kernel<128>, grid, dim3(128), 0, st, G, r, C, ldc, K, iters);
    else
        hipLaunchKernelGGL(mu_temporal_kernel<0>, grid, dim3(256), 0, st, G, r, C, ldc, K, iters);
    return check_launch("dnmf_mu_temporal");
}

int dnmf_mu_temporal_nbr(const float *G, const float *r, float *C, long ldc, int K, int T, int iters, const int *nbr,
                         int NN, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(G && r && C && nbr, DNMF_E_NULL, "dnmf_mu_temporal_nbr: NULL buffer");
    DNMF_REQUIRE(K > 0 && T > 0 && ldc >= T && iters >= 0, DNMF_E_SHAPE, "dnmf_mu_temporal_nbr: K=%d T=%d ldc=%ld iters=%d",
                 K, T, ldc, iters);
    DNMF_REQUIRE(K <= 256 && (NN == 8 || NN == 16 || NN == 32), DNMF_E_UNSUPPORTED,
                 "dnmf_mu_temporal_nbr: K=%d (<= 256), NN=%d (8, 16 or 32)", K, NN);
    if (iters == 0) return DNMF_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)T), block(K <= 64 ? 64 : (K <= 128 ? 128 : 256));
    if (NN == 8)
        hipLaunchKernelGGL(mu_temporal_nbr_kernel<8>, grid, block, 0, st, G, r, C, ldc, K, iters, nbr);
    else if (NN == 16)
        hipLaunchKernelGGL(mu_temporal_nbr_kernel<16>, grid, block, 0, st, G, r, C, ldc, K, iters, nbr);
    else
        hipLaunchKernelGGL(mu_temporal_nbr_kernel<32>, grid, block, 0, st, G, r, C, ldc, K, iters, nbr);
    return check_launch("dnmf_mu_temporal_nbr");
}

int dnmf_mu_temporal_slots(const float *slab, int nchunks, int nslot, const int *pair_slot, float *C, long ldc, int K,
                           int T, int iters, const int *nbr, int NN, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(slab && pair_slot && C && nbr, DNMF_E_NULL, "dnmf_mu_temporal_slots: NULL buffer");
    DNMF_REQUIRE(K > 0 && T > 0 && ldc >= T && iters >= 0 && nchunks > 0 && nslot > K, DNMF_E_SHAPE,
                 "dnmf_mu_temporal_slots: K=%d T=%d ldc=%ld iters=%d nchunks=%d nslot=%d", K, T, ldc, iters, nchunks, nslot);
    DNMF_REQUIRE(K <= 256 && (NN == 8 || NN == 16 || NN == 32), DNMF_E_UNSUPPORTED,
                 "dnmf_mu_temporal_slots: K=%d (<= 256), NN=%d (8, 16 or 32)", K, NN);
    if (iters == 0) return DNMF_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)T), block(K <= 64 ? 64 : (K <= 128 ? 128 : 256));
    if (NN == 8)
        hipLaunchKernelGGL(mu_temporal_slots_kernel<8>, grid, block, 0, st, slab, nchunks, nslot, pair_slot, C, ldc, K, iters, nbr);
    else if (NN == 16)
        hipLaunchKernelGGL(mu_temporal_slots_kernel<16>, grid, block, 0, st, slab, nchunks, nslot, pair_slot, C, ldc, K, iters, nbr);
    else
        hipLaunchKernelGGL(mu_temporal_slots_kernel<32>, grid, block, 0, st, slab, nchunks, nslot, pair_slot, C, ldc, K, iters, nbr);
    return check_launch("dnmf_mu_temporal_slots");
}

int dnmf_mu_temporal_step(const float *G, const float *r, const double *Cin, double *Cout, long ldc, int K, int T,
                          double gamma, const double *c_left, const double *c_right, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(G && r && Cin && Cout, DNMF_E_NULL, "dnmf_mu_temporal_step: NULL buffer");
    DNMF_REQUIRE(Cin != Cout, DNMF_E_SHAPE, "dnmf_mu_temporal_step: Cin and Cout must be distinct buffers");
    DNMF_REQUIRE(K > 0 && T > 0 && ldc >= T, DNMF_E_SHAPE, "dnmf_mu_temporal_step: K=%d T=%d ldc=%ld", K, T, ldc);
    DNMF_REQUIRE(K <= 256, DNMF_E_UNSUPPORTED, "dnmf_mu_temporal_step: K=%d > 256 (not built yet)", K);
    const int block = K <= 64 ? 64 : (K <= 128 ? 128 : 256);
    hipLaunchKernelGGL(mu_temporal_step_kernel, dim3((unsigned)T), dim3(block), 0, (hipStream_t)stream, G, r, Cin, Cout,
                       ldc, K, T, gamma, c_left, c_right);
    return check_launch("dnmf_mu_temporal_step");
}

}  // extern "C"
