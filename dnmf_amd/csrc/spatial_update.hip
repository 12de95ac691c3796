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
// Group g holds the voxels p0 + 4 i + g (i = 0..15), so that a lane's four A operands are one 16-byte load:
//   A operand: lane l -> Y[t + (l>>4)][p0 + 4 (l&15) + g]      (a wave reads 256 consecutive bytes of each of 4 frames)
//   B operand: lane l -> C[16 b + (l&15)][t + (l>>4)]
//   D tile   : lane l -> A1[p0 + 4 (4 (l>>4) + r) + g][16 b + (l&15)]
// Operands are requested a whole iteration of 16 frames ahead (round 1: load -> wait -> 28 MFMAs per step of 4 frames
// with two waves per SIMD to cover it, and conditional loads that the compiler turned into branches with a wait
// each: 4.8 ms per 4000 frames at 512x512, K=100 against 1.5 ms of fp32 matrix work).
template <int NB>
__global__ __launch_bounds__(256) void spatial_accum_kernel(const float *__restrict__ Y, long ldy,
                                                            const int *__restrict__ frame_ids,
                                                            const float *__restrict__ C, long ldc,
                                                            const float *__restrict__ Ct, long ldct,
                                                            const int *__restrict__ times, int T, long P, int K,
                                                            float *__restrict__ A1, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long p0 = ((long)blockIdx.x * 4 + wave) * 64;
    if (p0 >= P) return;
    const int ci = lane & 15, q = lane >> 4;
    // 16-byte loads when every row of Y allows them
    const bool wide = p0 + 64 <= P && (ldy & 3) == 0 && ((size_t)Y & 15) == 0;
    f32x4 acc[4][NB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // An iteration covers KS MFMA steps = 4 KS frames.  Its operands were requested one iteration earlier (and the frame
    // ids / trace columns they are addressed with one iteration before that): with two waves per SIMD the 28 KS MFMAs of
    // the two waves (~7 000 cycles) are what a request has to arrive, which covers an HBM round trip.
    constexpr int KS = 4;
    struct Rows {
        int fid[KS], col[KS];   // row of Y, column of C of this lane's frame in step s (clamped to the last frame)
    };
    auto rows_of = [&](int t0) {
        Rows r;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int tc = min(t0 + 4 * s + q, T - 1);
            r.fid[s] = frame_ids ? frame_ids[tc] : tc;
            r.col[s] = times ? times[tc] : tc;
        }
        return r;
    };
    // Loads without branches and without a use next to them: clamped addresses, raw values.  What must not count is
    // zeroed on the B side when the values become current (a frame past the end, a trace column past K); voxels past P
    // are never stored.
    auto fetch = [&](const Rows &r, f32x4 (&a)[KS], float (&bq)[KS][NB]) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float *yrow = Y + (long)r.fid[s] * ldy;
            if (wide) {
                a[s] = *reinterpret_cast<const f32x4 *>(yrow + p0 + 4 * ci);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long p = p0 + 4 * ci + g;
                    a[s][g] = yrow[p < P ? p : P - 1];
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int kc = min(16 * b + ci, K - 1);
                // frame-major traces: the 16 lanes of a frame read one 64-byte run; else 16 rows, 4 bytes of each
                bq[s][b] = Ct ? Ct[(long)r.col[s] * ldct + kc] : C[(long)kc * ldc + r.col[s]];
            }
        }
    };
    auto masked = [&](int t0, const float (&raw)[KS][NB], float (&bq)[KS][NB]) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int b = 0; b < NB; ++b) bq[s][b] = (t0 + 4 * s + q < T && 16 * b + ci < K) ? raw[s][b] : 0.0f;
    };
    f32x4 a_cur[KS], a_nxt[KS];
    float b_cur[KS][NB], b_nxt[KS][NB];
    Rows r_nxt = rows_of(0);
    fetch(r_nxt, a_cur, b_nxt);
    masked(0, b_nxt, b_cur);
    r_nxt = rows_of(4 * KS);
    for (int t0 = 0; t0 < T; t0 += 4 * KS) {
        fetch(r_nxt, a_nxt, b_nxt);
        const Rows r_nn = rows_of(t0 + 8 * KS);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s][g], b_cur[s][b], acc[g][b], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = a_nxt[s];
        masked(t0 + 4 * KS, b_nxt, b_cur);
        r_nxt = r_nn;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long p = p0 + 4 * (4 * q + r) + g;
                const int k = 16 * b + ci;
                if (p < P && k < K) {
                    float *dst = A1 + p * K + k;
                    *dst = accumulate ? *dst + acc[g][b][r] : acc[g][b][r];
                }
            }
}

// C_s[k][l] = sum_t C[k,t] C[l,t] in float64.  One block per (k, four l): the threads stride over the frames (both rows
// read in runs), wave sums by shuffles in a fixed order, the four waves through LDS.  (Round 1 had one block
// per row k walk over all l with a block reduction each: 100 blocks busy for 5 ms at K = 100, T = 4000.)
constexpr int TG_L = 4;
__global__ __launch_bounds__(256) void trace_gram_kernel(const float *__restrict__ C, long ldc,
                                                         const int *__restrict__ times, int T, int K,
                                                         float *__restrict__ Cs, int accumulate) {
    __shared__ double red[4][TG_L];
    const int k = blockIdx.x, l0 = blockIdx.y * TG_L;
    double s[TG_L];
#pragma unroll
    for (int j = 0; j < TG_L; ++j) s[j] = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const long c = times ? times[t] : t;
        const double ck = (double)C[(long)k * ldc + c];
#pragma unroll
        for (int j = 0; j < TG_L; ++j) s[j] += ck * (double)C[(long)min(l0 + j, K - 1) * ldc + c];
    }
#pragma unroll
    for (int j = 0; j < TG_L; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < TG_L && l0 + (int)threadIdx.x < K) {
        const int j = threadIdx.x;
        const double tot = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
        float *dst = Cs + (long)k * K + l0 + j;
        *dst = (accumulate ? *dst : 0.0f) + (float)tot;
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
static void launch_accum(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const float *Ct, long ldct,
                         const int *times, int T, long P, int K, float *A1, int accumulate, hipStream_t st) {
    const unsigned nwg = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL((spatial_accum_kernel<NB>), dim3(nwg), dim3(256), 0, st, Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P,
                       K, A1, accumulate);
}

}  // namespace dnmf

extern "C" {

int dnmf_spatial_accum(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const float *Ct, long ldct,
                       const int *times, int T, long P, int K, float *A1, float *Cs, int accumulate, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Y && C && A1 && Cs, DNMF_E_NULL, "dnmf_spatial_accum: NULL buffer");
    DNMF_REQUIRE(!Ct || ldct >= K, DNMF_E_SHAPE, "dnmf_spatial_accum: ldct=%ld < K=%d", ldct, K);
    DNMF_REQUIRE(T > 0 && P > 0 && K > 0 && ldy >= P && ldc > 0, DNMF_E_SHAPE,
                 "dnmf_spatial_accum: T=%d P=%ld K=%d ldy=%ld ldc=%ld", T, P, K, ldy, ldc);
    DNMF_REQUIRE(K <= 128, DNMF_E_UNSUPPORTED, "dnmf_spatial_accum: K=%d > 128 (not built yet)", K);
    hipStream_t st = (hipStream_t)stream;
    switch ((K + 15) / 16) {
        case 1: launch_accum<1>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 2: launch_accum<2>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 3: launch_accum<3>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 4: launch_accum<4>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 5: launch_accum<5>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 6: launch_accum<6>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        case 7: launch_accum<7>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
        default: launch_accum<8>(Y, ldy, frame_ids, C, ldc, Ct, ldct, times, T, P, K, A1, accumulate, st); break;
    }
    hipLaunchKernelGGL(trace_gram_kernel, dim3((unsigned)K, (unsigned)((K + TG_L - 1) / TG_L)), dim3(256), 0, st, C, ldc, times, T, K,
                       Cs, accumulate);
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
