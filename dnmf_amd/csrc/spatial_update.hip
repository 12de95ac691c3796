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


// ---- list form of the footprint update (compact footprints) -------------------------------------------------------
// K6 multiplies A1[p,k] = sum_t Y_i[t,p] C[k,t] by A[p,k], and a multiplicative update keeps every zero: outside the
// non-zero box of footprint k the product -- all that is ever used of A1 -- is an exact zero whatever A1 is.  The dense
// K5 above spends 2 P K T flops (2.4 10^11 at 512x512x4000, K = 100) where ~1.4 boxes cover a voxel.  Here a tile of
// 4 x-rows x 64 positions of the (y,z) plane lists the neurons whose box meets it (a static list per footprint version,
// 3 on average) and the sums exist only for (tile, listed neuron) pairs, in a compact buffer A1c of `total` floats: entry
// (tile q, list slot i, lane, v) at tile_off[q] + 256 i + 4 lane + v, lane = 16 row + (position / 4), i.e. 1 KiB runs.
// The kernel is bound by reading the registered video once; the buffer the ranks all-reduce shrinks from P K floats
// (105 MB) to `total` (~3 P: 3 MB).  K6's list form then reads the footprints of a tile's list (the only non-zeros of
// its rows) and evaluates the same expression, the denominator's sum over the listed neurons in ascending order: the
// terms it skips are exact zeros, so it equals the dense K6 on the same A1 bit for bit.
constexpr int SL_MAXL = 32;      // neurons per tile list
constexpr int SL_NG = 8;         // neurons accumulated together (registers)

struct SlGeom {
    int X, Y, Z, YZ, nu, ntx, ntiles;
};
static SlGeom sl_geom(int X, int Y, int Z) {
    SlGeom g;
    g.X = X, g.Y = Y, g.Z = Z, g.YZ = Y * Z, g.nu = (g.YZ + 63) / 64, g.ntx = (X + 3) / 4, g.ntiles = g.ntx * g.nu;
    return g;
}

// one wave per tile: its list (ascending neuron index) and length; -1 (and *overflow += 1) beyond SL_MAXL
__global__ __launch_bounds__(256) void sl_lists_kernel(const int *__restrict__ bbox, int K, SlGeom g, int *__restrict__ tile_n,
                                                       int *__restrict__ tile_list, int *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g.ntiles) return;
    const int qu = q % g.nu, qx = q / g.nu;
    const int xlo = 4 * qx, xhi = min(4 * qx + 3, g.X - 1);
    const int ylo = (64 * qu) / g.Z, yhi = min(64 * qu + 63, g.YZ - 1) / g.Z;
    int n = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < K) {
            const int *bb = bbox + k * 6;
            hit = bb[0] <= xhi && bb[1] >= xlo && bb[2] <= yhi && bb[3] >= ylo && bb[4] <= bb[5];
        }
        const unsigned long long m = __ballot(hit);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (hit && n + before < SL_MAXL) tile_list[(long)q * SL_MAXL + n + before] = k;
        n += __builtin_popcountll(m);
    }
    if (lane == 0) {
        tile_n[q] = n <= SL_MAXL ? n : -1;
        if (n > SL_MAXL) atomicAdd(overflow, 1);
    }
}

// tile_off[q] = 256 * sum_{q' < q} tile_n[q'] (one block; tile_off[ntiles] = total, -1 when a list overflowed)
__global__ __launch_bounds__(256) void sl_scan_kernel(const int *__restrict__ tile_n, int ntiles, int *__restrict__ tile_off,
                                                      const int *__restrict__ overflow) {
    __shared__ long part[256];
    const int per = (ntiles + 255) / 256;
    const int a = threadIdx.x * per, b = min(a + per, ntiles);
    long s = 0;
    for (int q = a; q < b; ++q) s += max(tile_n[q], 0);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long acc = 0;
        for (int i = 0; i < 256; ++i) {
            const long v = part[i];
            part[i] = acc, acc += v;
        }
    }
    __syncthreads();
    long run = part[threadIdx.x];
    for (int q = a; q < b; ++q) tile_off[q] = (int)(256 * run), run += max(tile_n[q], 0);
    if (threadIdx.x == 255) tile_off[ntiles] = *overflow ? -1 : (int)(256 * run);   // (thread 255's range ends at ntiles)
}

typedef float sl_f4 __attribute__((ext_vector_type(4)));

// One wave per (tile, split of the frames): partial sums of A1 for the tile's listed neurons over frames [t0, t1).
template <bool ALIGNED>
__global__ __launch_bounds__(256) void spatial_accum_lists_kernel(const float *__restrict__ Y, long ldy, const int *__restrict__ frame_ids,
                                                                  const float *__restrict__ C, long ldc, const int *__restrict__ times,
                                                                  int T, SlGeom g, const int *__restrict__ tile_n,
                                                                  const int *__restrict__ tile_off, const int *__restrict__ tile_list,
                                                                  float *__restrict__ out, long out_split_stride, int frames_per_split) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g.ntiles) return;
    const int n = tile_n[q];
    if (n <= 0) return;
    const int split = blockIdx.y;
    const int t0 = split * frames_per_split, t1 = min(t0 + frames_per_split, T);
    const int qu = q % g.nu, qx = q / g.nu;
    const int x = 4 * qx + (lane >> 4), u = 64 * qu + 4 * (lane & 15);
    const bool row_in = x < g.X;
    bool in[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) in[v] = row_in && u + v < g.YZ;
    // voxels that can always be read (entries outside the volume are masked below); ALIGNED: YZ is a multiple of 4, so a
    // lane inside the plane has all four positions inside
    const long prow = (long)min(x, g.X - 1) * g.YZ;
    const long p = prow + min(u, max(g.YZ - 4, 0));
    int pu[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) pu[v] = min(u + v, g.YZ - 1);
    const int *lst = tile_list + (long)q * SL_MAXL;
    float *dst = out + (long)split * out_split_stride + tile_off[q] + 4 * lane;
    for (int g0 = 0; g0 < n; g0 += SL_NG) {
        int ks[SL_NG];
#pragma unroll
        for (int i = 0; i < SL_NG; ++i) ks[i] = g0 + i < n ? lst[g0 + i] : -1;
        sl_f4 acc[SL_NG];
#pragma unroll
        for (int i = 0; i < SL_NG; ++i) acc[i] = sl_f4{0.f, 0.f, 0.f, 0.f};
        for (int bb = t0; bb < t1; bb += 64) {
            // lane j fetches the traces of frame bb + j and the row that holds it; the run reads them as scalars
            const int bl = min(bb + lane, t1 - 1);
            const long tcol = times ? times[bl] : bl;
            const int myrow = frame_ids ? frame_ids[bl] : bl;
            float cv[SL_NG];
#pragma unroll
            for (int i = 0; i < SL_NG; ++i) cv[i] = ks[i] >= 0 ? C[(long)ks[i] * ldc + tcol] : 0.0f;
            const int nb = min(64, t1 - bb);
            for (int j0 = 0; j0 < nb; j0 += 4) {
                sl_f4 y[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {     // four frames requested together
                    const int j = min(j0 + jj, nb - 1);
                    const float *src = Y + (long)__builtin_amdgcn_readlane(myrow, j) * ldy;
                    if (ALIGNED) {
                        y[jj] = *reinterpret_cast<const sl_f4 *>(src + p);
                    } else {
                        src += prow;
                        y[jj] = sl_f4{src[pu[0]], src[pu[1]], src[pu[2]], src[pu[3]]};
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (j0 + jj >= nb) break;       // wave-uniform
#pragma unroll
                    for (int i = 0; i < SL_NG; ++i) {
                        const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv[i]), j0 + jj));
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[i][v] = fmaf(y[jj][v], c, acc[i][v]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SL_NG; ++i)
            if (g0 + i < n) {
                sl_f4 r = acc[i];
#pragma unroll
                for (int v = 0; v < 4; ++v) r[v] = in[v] ? r[v] : 0.0f;
                *reinterpret_cast<sl_f4 *>(dst + 256 * (g0 + i)) = r;
            }
    }
}

// A1c[e] = sum over the splits, in order
__global__ void sl_reduce_kernel(const float *__restrict__ part, long stride, int nsplit, long total, float *__restrict__ A1c) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float s = 0.0f;
    for (int i = 0; i < nsplit; ++i) s += part[(long)i * stride + e];
    A1c[e] = s;
}

// Phase 1 of the list form of K6: A1c[entry] <- A A1 / (sum_l A[.,l] Cs[l,k] + gamma D + 1e-32) for every (tile, listed
// neuron, voxel); phase 2 scatters the entries into A (P,K).  Two phases because the update is in place and a tile's rows
// are read by all of its entries.  `At` = the neuron-major halo-layout copy of A (dnmf_pack_footprints_lists).
__global__ __launch_bounds__(256) void mu_spatial_lists_kernel(const float *__restrict__ At, long Pp, int rowf, float *__restrict__ A1c,
                                                               const float *__restrict__ Cs, const float *__restrict__ D, float gamma,
                                                               int K, SlGeom g, const int *__restrict__ tile_n,
                                                               const int *__restrict__ tile_off, const int *__restrict__ tile_list) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g.ntiles) return;
    const int n = tile_n[q];
    if (n <= 0) return;
    const int qu = q % g.nu, qx = q / g.nu;
    const int x = 4 * qx + (lane >> 4), u = 64 * qu + 4 * (lane & 15);
    const int *lst = tile_list + (long)q * SL_MAXL;
    float *ent = A1c + tile_off[q] + 4 * lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if (x >= g.X || u + v >= g.YZ) continue;
        const long hp = (long)(x + HALO) * rowf + (long)HALO * g.Z + u + v;      // halo index of the voxel
        const long p = (long)x * g.YZ + u + v;
        for (int i = 0; i < n; ++i) {
            const int ki = lst[i];
            float den = 0.0f;
            for (int j = 0; j < n; ++j) den = fmaf(At[(long)lst[j] * Pp + hp], Cs[(long)lst[j] * K + ki], den);
            if (D) den += gamma * D[p * K + ki];
            const float a = At[(long)ki * Pp + hp];
            ent[256 * i + v] = a * ent[256 * i + v] / (den + 1e-32f);
        }
    }
}

__global__ __launch_bounds__(256) void mu_spatial_scatter_kernel(float *__restrict__ A, const float *__restrict__ A1c, int K, SlGeom g,
                                                                 const int *__restrict__ tile_n, const int *__restrict__ tile_off,
                                                                 const int *__restrict__ tile_list) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g.ntiles) return;
    const int n = tile_n[q];
    if (n <= 0) return;
    const int qu = q % g.nu, qx = q / g.nu;
    const int x = 4 * qx + (lane >> 4), u = 64 * qu + 4 * (lane & 15);
    const int *lst = tile_list + (long)q * SL_MAXL;
    const float *ent = A1c + tile_off[q] + 4 * lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if (x >= g.X || u + v >= g.YZ) continue;
        const long p = (long)x * g.YZ + u + v;
        for (int i = 0; i < n; ++i) A[p * K + lst[i]] = ent[256 * i + v];
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


// ---- list form (see the kernels above) ----
long dnmf_spatial_lists_tiles(int X, int Y, int Z) {
    if (X <= 0 || Y <= 0 || Z <= 0) return 0;
    return dnmf::sl_geom(X, Y, Z).ntiles;
}

int dnmf_spatial_lists_setup(const int *bbox, int K, int X, int Y, int Z, int *tables, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(bbox && tables, DNMF_E_NULL, "dnmf_spatial_lists_setup: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0, DNMF_E_SHAPE, "dnmf_spatial_lists_setup: X=%d Y=%d Z=%d K=%d", X, Y, Z, K);
    const SlGeom g = sl_geom(X, Y, Z);
    DNMF_REQUIRE((long)X * Y * Z * SL_MAXL < (1L << 31), DNMF_E_UNSUPPORTED, "dnmf_spatial_lists_setup: volume too large for 32-bit offsets");
    int *tile_n = tables, *tile_off = tables + g.ntiles, *tile_list = tables + 2 * g.ntiles + 2;
    int *overflow = tables + 2 * g.ntiles + 1;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(overflow, 0, sizeof(int), st);
    DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_spatial_lists_setup: hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sl_lists_kernel, dim3((unsigned)((g.ntiles + 3) / 4)), dim3(256), 0, st, bbox, K, g, tile_n, tile_list, overflow);
    hipLaunchKernelGGL(sl_scan_kernel, dim3(1), dim3(256), 0, st, tile_n, g.ntiles, tile_off, overflow);
    return check_launch("dnmf_spatial_lists_setup");
}

static int sl_splits(int ntiles, int T) {
    int s = (8192 + ntiles - 1) / ntiles;     // >= 8192 waves
    s = s < 1 ? 1 : (s > 64 ? 64 : s);
    while (s > 1 && T / s < 64) --s;          // whole runs of 64 frames per wave
    return s;
}

size_t dnmf_spatial_accum_lists_workspace(int X, int Y, int Z, long total, int T) {
    if (X <= 0 || Y <= 0 || Z <= 0 || total <= 0 || T <= 0) return 0;
    const int s = sl_splits(dnmf::sl_geom(X, Y, Z).ntiles, T);
    return s > 1 ? (size_t)s * total * sizeof(float) : 0;
}

int dnmf_spatial_accum_lists(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const int *times, int T, int X,
                             int Yd, int Z, int K, const int *tables, long total, float *A1c, float *Cs, void *workspace,
                             size_t workspace_bytes, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Y && C && tables && A1c && Cs, DNMF_E_NULL, "dnmf_spatial_accum_lists: NULL buffer");
    DNMF_REQUIRE(T > 0 && X > 0 && Yd > 0 && Z > 0 && K > 0 && total > 0 && ldy >= (long)X * Yd * Z && ldc > 0, DNMF_E_SHAPE,
                 "dnmf_spatial_accum_lists: T=%d X=%d Y=%d Z=%d K=%d total=%ld ldy=%ld", T, X, Yd, Z, K, total, ldy);
    const SlGeom g = sl_geom(X, Yd, Z);
    DNMF_REQUIRE(g.YZ >= 4, DNMF_E_UNSUPPORTED, "dnmf_spatial_accum_lists: Y*Z = %d < 4", g.YZ);
    const int ns = sl_splits(g.ntiles, T);
    DNMF_REQUIRE(ns == 1 || (workspace && workspace_bytes >= (size_t)ns * total * sizeof(float)), DNMF_E_WORKSPACE,
                 "dnmf_spatial_accum_lists: workspace %zu < %zu bytes", workspace_bytes, (size_t)ns * total * sizeof(float));
    const int *tile_n = tables, *tile_off = tables + g.ntiles, *tile_list = tables + 2 * g.ntiles + 2;
    hipStream_t st = (hipStream_t)stream;
    const int fps = (T + ns - 1) / ns;
    float *out = ns > 1 ? static_cast<float *>(workspace) : A1c;
    const dim3 grid((unsigned)((g.ntiles + 3) / 4), (unsigned)ns);
    const bool aligned = (g.YZ & 3) == 0 && (ldy & 3) == 0 && ((size_t)Y & 15) == 0;
    // entries of tiles the kernel leaves (none listed) do not exist; every existing entry is written by every split
    if (aligned)
        hipLaunchKernelGGL((spatial_accum_lists_kernel<true>), grid, dim3(256), 0, st, Y, ldy, frame_ids, C, ldc, times, T, g, tile_n,
                           tile_off, tile_list, out, total, fps);
    else
        hipLaunchKernelGGL((spatial_accum_lists_kernel<false>), grid, dim3(256), 0, st, Y, ldy, frame_ids, C, ldc, times, T, g, tile_n,
                           tile_off, tile_list, out, total, fps);
    if (ns > 1) hipLaunchKernelGGL(sl_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total, ns, total, A1c);
    hipLaunchKernelGGL(trace_gram_kernel, dim3((unsigned)K, (unsigned)((K + TG_L - 1) / TG_L)), dim3(256), 0, st, C, ldc, times, T, K,
                       Cs, 0);
    return check_launch("dnmf_spatial_accum_lists");
}

int dnmf_mu_spatial_lists(float *A, const float *At, float *A1c, const float *Cs, const float *D, double gamma, int X, int Y, int Z,
                          int K, const int *tables, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && At && A1c && Cs && tables, DNMF_E_NULL, "dnmf_mu_spatial_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0, DNMF_E_SHAPE, "dnmf_mu_spatial_lists: X=%d Y=%d Z=%d K=%d", X, Y, Z, K);
    const SlGeom g = sl_geom(X, Y, Z);
    const HaloLayout hl = make_halo_layout(X, Y, Z);
    const int *tile_n = tables, *tile_off = tables + g.ntiles, *tile_list = tables + 2 * g.ntiles + 2;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((g.ntiles + 3) / 4));
    hipLaunchKernelGGL(mu_spatial_lists_kernel, grid, dim3(256), 0, st, At, hl.Pp, hl.rowf, A1c, Cs, D, (float)gamma, K, g, tile_n,
                       tile_off, tile_list);
    hipLaunchKernelGGL(mu_spatial_scatter_kernel, grid, dim3(256), 0, st, A, A1c, K, g, tile_n, tile_off, tile_list);
    return check_launch("dnmf_mu_spatial_lists");
}

}  // extern "C"
