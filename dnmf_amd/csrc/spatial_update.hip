// K5 / K6 -- multiplicative update of the (un-warped) footprints.
//
// Reference: DeformableNMF.update_spatial, Demix/dNMF.py:151-160
//   C_s = einsum('kt,pt->kp', C, C)            (:153)   K x K
//   A1  = einsum('mnt,kt->mnk', Y_i, C)        (:154)   (P x T).(T x K)  -- the large one, a sum over ALL frames
//   A2  = einsum('mnk,kp->mnp', A, C_s) (+ gamma*D)     (:155-158)
//   A   = A * A1 / (A2 + 1e-32)                (:159)
// K5 (dnmf_spatial_accum) produces A1 and C_s for the frames this process holds; with the T axis sharded the
// two buffers are summed over ranks (one RCCL all-reduce each, done by the caller) before K6
// (dnmf_mu_spatial) applies the ratio.  fp32 MFMA for A1 (v_mfma_f32_16x16x4_f32, frames on the reduction axis).
#include "common.hpp"

namespace dnmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave: 64 voxels (4 groups of 16) x NB blocks of 16 traces, reduction over all T frames, 4 per MFMA.
//   A operand: lane l -> Y[t + (l>>4)][p0 + 16 g + (l&15)]     (16 consecutive voxels of one frame = 64 B)
//   B operand: lane l -> C[16 b + (l&15)][t + (l>>4)]
//   D tile   : lane l -> A1[p0 + 16 g + 4 (l>>4) + r][16 b + (l&15)]
template <int NB>
__global__ __launch_bounds__(256) void spatial_accum_kernel(const float *__restrict__ Y, long ldy,
                                                            const int *__restrict__ frame_ids,
                                                            const float *__restrict__ C, long ldc,
                                                            const int *__restrict__ times, int T, long P, int K,
                                                            float *__restrict__ A1, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long p0 = ((long)blockIdx.x * 4 + wave) * 64;
    if (p0 >= P) return;
    const int ci = lane & 15, q = lane >> 4;
    f32x4 acc[4][NB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t0 = 0; t0 < T; t0 += 4) {
        const int tt = t0 + q;
        const bool live = tt < T;
        const int tc = live ? tt : T - 1;
        const float *yrow = Y + (long)(frame_ids ? frame_ids[tc] : tc) * ldy;
        const long ccol = times ? times[tc] : tc;
        float a[4], bq[NB];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long p = p0 + 16 * g + ci;
            a[g] = (live && p < P) ? yrow[p] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int k = 16 * b + ci;
            bq[b] = (live && k < K) ? C[(long)k * ldc + ccol] : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], bq[b], acc[g][b], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long p = p0 + 16 * g + 4 * q + r;
                const int k = 16 * b + ci;
                if (p < P && k < K) {
                    float *dst = A1 + p * K + k;
                    *dst = accumulate ? *dst + acc[g][b][r] : acc[g][b][r];
                }
            }
}

// C_s[k][l] = sum_t C[k,t] C[l,t]; one block per row k, fp32 partial sums per thread, fp64 combine
__global__ __launch_bounds__(256) void trace_gram_kernel(const float *__restrict__ C, long ldc,
                                                         const int *__restrict__ times, int T, int K,
                                                         float *__restrict__ Cs, int accumulate) {
    __shared__ double red[256];
    const int k = blockIdx.x;
    for (int l = 0; l < K; ++l) {
        double s = 0.0;
        for (int t = threadIdx.x; t < T; t += 256) {
            const long c = times ? times[t] : t;
            s += (double)C[(long)k * ldc + c] * (double)C[(long)l * ldc + c];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) Cs[(long)k * K + l] = (accumulate ? Cs[(long)k * K + l] : 0.0f) + (float)red[0];
        __syncthreads();
    }
}

// A[p][k] <- A[p][k] * A1[p][k] / (sum_l A[p][l] Cs[l][k] + gamma D[p][k] + 1e-32); 16 voxels per block,
// the footprint rows of the block staged in LDS (the update is in place: all rows are read before any write)
__global__ __launch_bounds__(256) void mu_spatial_kernel(float *__restrict__ A, const float *__restrict__ A1,
                                                         const float *__restrict__ Cs, const float *__restrict__ D,
                                                         float gamma, long P, int K) {
    extern __shared__ float rows[];  // 16 x K
    const long p0 = (long)blockIdx.x * 16;
    const int nrow = (int)((P - p0) < 16 ? (P - p0) : 16);
    for (int i = threadIdx.x; i < nrow * K; i += blockDim.x) rows[i] = A[p0 * K + i];
    __syncthreads();
    for (int i = threadIdx.x; i < nrow * K; i += blockDim.x) {
        const int r = i / K, k = i - r * K;
        const float *a = rows + r * K;
        float den = 0.0f;
        for (int l = 0; l < K; ++l) den = fmaf(a[l], Cs[(long)l * K + k], den);
        if (D) den += gamma * D[(p0 + r) * K + k];
        A[p0 * K + i] = a[k] * A1[p0 * K + i] / (den + 1e-32f);
    }
}

template <int NB>
static void launch_accum(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const int *times,
                         int T, long P, int K, float *A1, int accumulate, hipStream_t st) {
    const unsigned nwg = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL((spatial_accum_kernel<NB>), dim3(nwg), dim3(256), 0, st, Y, ldy, frame_ids, C, ldc, times, T, P,
                       K, A1, accumulate);
}

}  // namespace dnmf

extern "C" {

int dnmf_spatial_accum(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const int *times,
                       int T, long P, int K, float *A1, float *Cs, int accumulate, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Y && C && A1 && Cs, DNMF_E_NULL, "dnmf_spatial_accum: NULL buffer");
    DNMF_REQUIRE(T > 0 && P > 0 && K > 0 && ldy >= P && ldc > 0, DNMF_E_SHAPE,
                 "dnmf_spatial_accum: T=%d P=%ld K=%d ldy=%ld ldc=%ld", T, P, K, ldy, ldc);
    DNMF_REQUIRE(K <= 128, DNMF_E_UNSUPPORTED, "dnmf_spatial_accum: K=%d > 128 (not built yet)", K);
    hipStream_t st = (hipStream_t)stream;
    switch ((K + 15) / 16) {
        case 1: launch_accum<1>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 2: launch_accum<2>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 3: launch_accum<3>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 4: launch_accum<4>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 5: launch_accum<5>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 6: launch_accum<6>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        case 7: launch_accum<7>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
        default: launch_accum<8>(Y, ldy, frame_ids, C, ldc, times, T, P, K, A1, accumulate, st); break;
    }
    hipLaunchKernelGGL(trace_gram_kernel, dim3((unsigned)K), dim3(256), 0, st, C, ldc, times, T, K, Cs, accumulate);
    return check_launch("dnmf_spatial_accum");
}

int dnmf_mu_spatial(float *A, const float *A1, const float *Cs, const float *D, double gamma, long P, int K,
                    dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && A1 && Cs, DNMF_E_NULL, "dnmf_mu_spatial: NULL buffer");
    DNMF_REQUIRE(P > 0 && K > 0, DNMF_E_SHAPE, "dnmf_mu_spatial: P=%ld K=%d", P, K);
    DNMF_REQUIRE((size_t)K * 16 * sizeof(float) <= 64 * 1024, DNMF_E_UNSUPPORTED, "dnmf_mu_spatial: K=%d too large", K);
    hipLaunchKernelGGL(mu_spatial_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), (size_t)K * 16 * sizeof(float),
                       (hipStream_t)stream, A, A1, Cs, D, (float)gamma, P, K);
    return check_launch("dnmf_mu_spatial");
}

}  // extern "C"
