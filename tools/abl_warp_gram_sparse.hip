// K3s -- the Gram kernel with exact-zero block skipping.
//
// Same contraction as warp_gram_rhs.hip (G_t = A_t^T A_t, r_t = A_t^T y_t; reference Demix/dNMF.py:141-142), for
// footprints that are exactly zero over most of the volume: the reference's Gaussians underflow to 0 in fp32
// beyond ~30 px, and a multiplicative update keeps every zero a zero.  Products with an exact zero add nothing,
// so leaving them out changes no sum.
//
// Neurons are ordered along a space-filling curve (host) and cut into blocks of 16 channels; every footprint
// row carries a bit mask of the blocks in which it has a non-zero.  For each pass of 64 voxels a wave ORs the
// masks of all source rows it is about to gather (wave-uniform set S), gathers and blends ONLY the blocks in S
// for the 8 k-steps of each half pass (one set per half), and issues the MFMAs of tile (bi,bj) only if both
// blocks are in S; a half pass with S empty costs its share of the coordinate pass and nothing else.  Accumulators stay statically indexed: the skips are
// scalar branches around fully unrolled code.  The right-hand side is accumulated on the vector ALU.
#include "../dnmf_amd/csrc/common.hpp"

#include <type_traits>

namespace dnmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KS_SS = 64;          // voxels per pass
constexpr int KS_NKS = KS_SS / 4;  // k-steps per pass
constexpr int KS_HALF = KS_NKS / 2;  // k-steps per half pass (block sets are per half)

__host__ __device__ constexpr int sp_tile_index(int NB, int bi, int bj) { return bi * NB - bi * (bi - 1) / 2 + (bj - bi); }

// Aps[p][c] = c < K ? A[p][order[c]] : 0 ;  row_mask[p] bit b = any(Aps[p][16b .. 16b+15] != 0)
__global__ __launch_bounds__(256) void pack_sparse_kernel(const float *__restrict__ A, long P, int K,
                                                          const int *__restrict__ order, float *__restrict__ Aps,
                                                          int Ks, unsigned char *__restrict__ row_mask) {
    const int sub = threadIdx.x & 15;                                  // 16 threads per row
    const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= P) return;  // rows are handed out in groups of 16 lanes: a group leaves together
    unsigned m = 0;
    for (int b = 0; b < Ks / 16; ++b) {
        const int c = 16 * b + sub;
        const float v = c < K ? A[p * K + order[c]] : 0.0f;
        Aps[p * Ks + c] = v;
        unsigned nz = v != 0.0f ? 1u : 0u;
        nz |= __shfl_xor(nz, 1, 16);
        nz |= __shfl_xor(nz, 2, 16);
        nz |= __shfl_xor(nz, 4, 16);
        nz |= __shfl_xor(nz, 8, 16);
        m |= nz << b;
    }
    if (sub == 0) row_mask[p] = (unsigned char)m;
}

struct SparseParams {
    const float *Aps;
    const unsigned char *row_mask;
    int Ks, K;
    Volume vol;
    const float *beta;
    int T;
    const int *times;
    int B;
    const float *frames;
    long ldf;
    const int *frame_ids;
    float *slab;  // (B, nchunks, NT*256 + 128)
    int nchunks;
    long chunk_len;    // passes (patches of 64 voxels) per chunk
    // a pass is a compact patch of 2^lgx x 2^lgy x 2^lgz = 64 voxels (8x8 for Z == 1), not a run of 64 voxels: fewer
    // footprints reach into a compact patch, so fewer blocks and tiles are active per pass
    int lgy, lgz, npy, npz;
    long npatch;
    unsigned long long *counters;  // optional: [0] += MFMAs issued, [1] += (active block, k-step) blends
};

template <int NB, int NTAP>
__global__ __launch_bounds__(256, NTAP == 4 ? 2 : 1) void warp_gram_sparse_kernel(SparseParams p) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NQ = NTAP / 4;
    constexpr int SLAB = NT * 256 + 128;
    __shared__ u32x4 s_row[4][NQ][KS_SS];
    __shared__ f32x4 s_w[4][NQ][KS_SS];
    __shared__ float s_y[4][KS_SS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)p.nchunks * p.B) return;
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const char *__restrict__ Ab = reinterpret_cast<const char *>(p.Aps);
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const unsigned row_bytes = (unsigned)p.Ks * 4u;

    float bt[30];
    load_beta(p.beta, p.T, t, bt);
    const int ci = lane & 15, vq = lane >> 4;
    const unsigned lane_off = 4u * ci;

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float racc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) racc[i] = 0.0f;

    const long q_begin = (long)chunk * p.chunk_len;
    const long q_end = q_begin + p.chunk_len < p.npatch ? q_begin + p.chunk_len : p.npatch;
    const int nss = (int)(q_end - q_begin);
    // patch origin of the first pass (wave-uniform), advanced with carries; lane -> voxel inside the patch
    const int lgy = p.lgy, lgz = p.lgz;
    int pz = (int)(q_begin % p.npz), py = (int)((q_begin / p.npz) % p.npy), px = (int)(q_begin / ((long)p.npz * p.npy));
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & ((1 << lgy) - 1), lx = lane >> (lgz + lgy);
    const int lgx = 6 - lgy - lgz;
    unsigned n_tiles = 0, n_blocks = 0;  // (tile, half pass) and (block, half pass) pairs executed

    for (int s = 0; s < nss; ++s) {
        // ---- coordinate pass + block set of the pass ------------------------------------------------------
        unsigned m = 0;
        {
            const int x = (px << lgx) + lx, y = (py << lgy) + ly, z = (pz << lgz) + lz;
            unsigned rows[NTAP], voxs[NTAP];
            float w[NTAP];
            float yv = 0.0f;
#pragma unroll
            for (int c = 0; c < NTAP; ++c) rows[c] = 0u, voxs[c] = 0u, w[c] = 0.0f;
            if (x < vol.X && y < vol.Y && z < vol.Z) {
                const Sample sm = make_sample_t<(NTAP == 8)>(bt, vol, x, y, z);
                make_taps<NTAP>(sm, vol, w, voxs);
#pragma unroll
                for (int c = 0; c < NTAP; ++c) rows[c] = voxs[c] * row_bytes;
                yv = yb[((long)x * vol.Y + y) * vol.Z + z];
                // all mask bytes requested together (one round trip); a tap without weight contributes no block
                unsigned mk[NTAP];
#pragma unroll
                for (int c = 0; c < NTAP; ++c) mk[c] = p.row_mask[voxs[c]];
#pragma unroll
                for (int c = 0; c < NTAP; ++c) m |= (w[c] != 0.0f) ? mk[c] : 0u;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                s_row[wave][q][lane] = u32x4{rows[4 * q], rows[4 * q + 1], rows[4 * q + 2], rows[4 * q + 3]};
                s_w[wave][q][lane] = f32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            }
            s_y[wave][lane] = yv;
        }
        // wave-uniform block sets of the two half passes (lanes 0-31 hold voxels 0-31 = k-steps 0-7)
        unsigned S2[2] = {0u, 0u};
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const unsigned long long bal = __ballot((m >> bb) & 1u);
            S2[0] |= ((unsigned)(bal & 0xffffffffull) != 0u ? 1u : 0u) << bb;
            S2[1] |= ((unsigned)(bal >> 32) != 0u ? 1u : 0u) << bb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const unsigned S = __builtin_amdgcn_readfirstlane(S2[h]);
            if (S == 0) continue;
            const int ks0 = h * KS_HALF;
            const int na = __builtin_popcount(S);
            n_blocks += na, n_tiles += na * (na + 1) / 2;
            // ---- gather + blend the active blocks for the 8 k-steps of this half -----------------------------
            float frag[NB][KS_HALF];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                if ((S >> bb) & 1u) {
                    const unsigned boff = 64u * bb + lane_off;
#pragma unroll
                    for (int ks = 0; ks < KS_HALF; ++ks) {
                        const int vi = (ks0 + ks) * 4 + vq;
                        float v = 0.0f;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const u32x4 rr = s_row[wave][q][vi];
                            const f32x4 ww = s_w[wave][q][vi];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v = fmaf(*reinterpret_cast<const float *>(Ab + (size_t)(rr[e] + boff)), ww[e], v);
                        }
                        frag[bb][ks] = v;
                        racc[bb] = fmaf(v, s_y[wave][vi], racc[bb]);
                    }
                }
            }
            // ---- MFMAs of the tiles whose two blocks are both active -----------------------------------------
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if ((S >> bi) & 1u) {
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj) {
                        if ((S >> bj) & 1u) {
                            const int idx = sp_tile_index(NB, bi, bj);
#pragma unroll
                            for (int ks = 0; ks < KS_HALF; ++ks)
                                acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[bi][ks], frag[bj][ks], acc[idx], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (++pz == p.npz) {
            pz = 0;
            if (++py == p.npy) py = 0, ++px;
        }
        // the next coordinate pass overwrites the records: keep it behind this pass's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (p.counters && lane == 0) {
        atomicAdd(p.counters + 0, (unsigned long long)n_tiles * KS_HALF);
        atomicAdd(p.counters + 1, (unsigned long long)n_blocks * KS_HALF);
    }
    float *out = p.slab + ((long)b * p.nchunks + chunk) * SLAB;
    f32x4 *out4 = reinterpret_cast<f32x4 *>(out) + lane;
#pragma unroll
    for (int i = 0; i < NT; ++i) out4[(long)i * 64] = acc[i];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        float v = racc[bb];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (vq == 0) out[NT * 256 + 16 * bb + ci] = v;
    }
}

// ---- K3s with a local block table ----------------------------------------------------------------------
// Same skipping rule as warp_gram_sparse_kernel, but a wave keeps accumulators only for the blocks it is currently
// working with: NL = 4 table slots (global block ids) and the 10 tiles among them -- 40 accumulator registers
// instead of 4*NT -- so four waves fit on a SIMD and hide each other's gather latency.  When a block outside the
// table shows up, a slot whose block is not needed by the current half pass is flushed: its tiles are added
// (read-modify-write, the region is private to the work item) into the item's slab, which holds the FULL NB x NB
// tile grid; a tile lands at (table[si], table[sj]) in whichever order the two blocks sit in the table, and the
// finish kernel adds the transposed partner.  A half pass with more than NL active blocks (dense spots) is done
// in rounds over pairs of 2-block chunks.
template <int NB, int NTAP>
__global__ __launch_bounds__(256, 3) void warp_gram_lt_kernel(SparseParams p) {
    constexpr int NL = 4;
    constexpr int NQ = NTAP / 4;
    constexpr int SLAB = NB * NB * 256 + 128;
    constexpr int EMPTY = 15;
    constexpr int KP = 8;            // k-steps per part: block sets are formed per half pass (32 voxels)
    constexpr int NPART = KS_NKS / KP;
    static_assert(NPART == 2, "the block sets are formed for lanes 0-31 / 32-63");
    __shared__ u32x4 s_row[4][NQ][KS_SS];
    __shared__ f32x4 s_w[4][NQ][KS_SS];
    __shared__ float s_y[4][KS_SS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)p.nchunks * p.B) return;
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const char *__restrict__ Ab = reinterpret_cast<const char *>(p.Aps);
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const unsigned row_bytes = (unsigned)p.Ks * 4u;
    float *slab = p.slab + ((long)b * p.nchunks + chunk) * SLAB;  // zeroed by the launch function
    f32x4 *slab4 = reinterpret_cast<f32x4 *>(slab);

    float bt[30];
    load_beta(p.beta, p.T, t, bt);
    const int ci = lane & 15, vq = lane >> 4;
    const unsigned lane_off = 4u * ci;

    f32x4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float racc[NL] = {0.f, 0.f, 0.f, 0.f};
    int tab[NL] = {EMPTY, EMPTY, EMPTY, EMPTY};  // wave-uniform: global block in each slot
    unsigned have = 0, dirty = 0;                // blocks in the table; local tiles with something to flush
    unsigned n_tiles = 0, n_blocks = 0;

    auto lt = [](int si, int sj) { return si * 4 - si * (si - 1) / 2 + (sj - si); };

    // add slot s's tiles and right-hand side into the slab, free the slot
    auto flush_slot = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int si = s < u ? s : u, sj = s < u ? u : s;
            const int l = lt(si, sj);
            if ((dirty >> l) & 1u) {
                f32x4 *ptr = slab4 + ((long)(tab[si] * NB + tab[sj]) * 64 + lane);
                *ptr = *ptr + acc[l];
                acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
                dirty &= ~(1u << l);
            }
        }
        float v = racc[s];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (vq == 0) slab[NB * NB * 256 + 16 * tab[s] + ci] += v;
        racc[s] = 0.0f;
        have &= ~(1u << tab[s]);
        tab[s] = EMPTY;
    };
    auto flush_dyn = [&](int s) {
        if (s == 0) flush_slot(std::integral_constant<int, 0>{});
        else if (s == 1) flush_slot(std::integral_constant<int, 1>{});
        else if (s == 2) flush_slot(std::integral_constant<int, 2>{});
        else flush_slot(std::integral_constant<int, 3>{});
    };
    auto flush_all = [&]() {
        if (tab[0] != EMPTY) flush_slot(std::integral_constant<int, 0>{});
        if (tab[1] != EMPTY) flush_slot(std::integral_constant<int, 1>{});
        if (tab[2] != EMPTY) flush_slot(std::integral_constant<int, 2>{});
        if (tab[3] != EMPTY) flush_slot(std::integral_constant<int, 3>{});
    };
    auto set_slot = [&](int s, int blk) {
        if (s == 0) tab[0] = blk;
        else if (s == 1) tab[1] = blk;
        else if (s == 2) tab[2] = blk;
        else tab[3] = blk;
    };

    // gather the slots in Lg for the 8 k-steps starting at ks0 (right-hand side for the slots in Lr), then the MFMAs
    // of the local tiles in tm
    auto process = [&](int ks0, unsigned Lg, unsigned Lr, unsigned tm) {
        float frag[NL][KP];
#pragma unroll
        for (int sl = 0; sl < NL; ++sl) {
            if ((Lg >> sl) & 1u) {
                const unsigned boff = 64u * (unsigned)tab[sl] + lane_off;
                float rsum = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) {
                    const int vi = (ks0 + ks) * 4 + vq;
                    float v = 0.0f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const u32x4 rr = s_row[wave][q][vi];
                        const f32x4 ww = s_w[wave][q][vi];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v = fmaf(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(&s_w[wave][0][0]) + ((rr[e] + boff) & 1020u)), ww[e], v);
                    }
                    frag[sl][ks] = v;
                    rsum = fmaf(v, s_y[wave][vi], rsum);
                }
                if ((Lr >> sl) & 1u) racc[sl] += rsum;
            }
        }
#pragma unroll
        for (int si = 0; si < NL; ++si) {
#pragma unroll
            for (int sj = si; sj < NL; ++sj) {
                const int l = lt(si, sj);
                if ((tm >> l) & 1u) {
#pragma unroll
                    for (int ks = 0; ks < KP; ++ks)
                        acc[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[si][ks], frag[sj][ks], acc[l], 0, 0, 0);
                }
            }
        }
        dirty |= tm;
        n_blocks += __builtin_popcount(Lg), n_tiles += __builtin_popcount(tm);
    };
    auto pairs_within = [&](unsigned L) {
        unsigned tm = 0;
#pragma unroll
        for (int si = 0; si < NL; ++si)
#pragma unroll
            for (int sj = si; sj < NL; ++sj) tm |= ((L >> si) & (L >> sj) & 1u) << lt(si, sj);
        return tm;
    };

    const long q_begin = (long)chunk * p.chunk_len;
    const long q_end = q_begin + p.chunk_len < p.npatch ? q_begin + p.chunk_len : p.npatch;
    const int nss = (int)(q_end - q_begin);
    const int lgy = p.lgy, lgz = p.lgz;
    int pz = (int)(q_begin % p.npz), py = (int)((q_begin / p.npz) % p.npy), px = (int)(q_begin / ((long)p.npz * p.npy));
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & ((1 << lgy) - 1), lx = lane >> (lgz + lgy);
    const int lgx = 6 - lgy - lgz;

    for (int s = 0; s < nss; ++s) {
        // ---- coordinate pass + block sets (as in warp_gram_sparse_kernel) -------------------------------------
        unsigned m = 0;
        {
            const int x = (px << lgx) + lx, y = (py << lgy) + ly, z = (pz << lgz) + lz;
            unsigned rows[NTAP], voxs[NTAP];
            float w[NTAP];
            float yv = 0.0f;
#pragma unroll
            for (int c = 0; c < NTAP; ++c) rows[c] = 0u, voxs[c] = 0u, w[c] = 0.0f;
            if (x < vol.X && y < vol.Y && z < vol.Z) {
                const Sample sm = make_sample_t<(NTAP == 8)>(bt, vol, x, y, z);
                make_taps<NTAP>(sm, vol, w, voxs);
#pragma unroll
                for (int c = 0; c < NTAP; ++c) rows[c] = voxs[c] * row_bytes;
                yv = yb[((long)x * vol.Y + y) * vol.Z + z];
                unsigned mk[NTAP];
#pragma unroll
                for (int c = 0; c < NTAP; ++c) mk[c] = p.row_mask[voxs[c]];
#pragma unroll
                for (int c = 0; c < NTAP; ++c) m |= (w[c] != 0.0f) ? mk[c] : 0u;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                s_row[wave][q][lane] = u32x4{rows[4 * q], rows[4 * q + 1], rows[4 * q + 2], rows[4 * q + 3]};
                s_w[wave][q][lane] = f32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            }
            s_y[wave][lane] = yv;
        }
        // wave-uniform block sets of the two half passes: OR of m over lanes 0-31 and 32-63 (rows of 16 lanes first
        // with shifted DPP reads, then row 0 into row 1 and row 2 into row 3), read from lanes 31 and 63
        unsigned Sw;
        {
            int v = (int)m;
            v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
            v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
            v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
            v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
            v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1 and 3
            Sw = (unsigned)__builtin_amdgcn_readlane(v, 31) | ((unsigned)__builtin_amdgcn_readlane(v, 63) << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#pragma unroll 1
        for (int h = 0; h < NPART; ++h) {
            const unsigned S = (Sw >> (8 * h)) & 0xffu;
            if (S == 0) continue;
            const int ks0 = h * KP;
            if (__builtin_popcount(S) <= NL) {
                // make room for the blocks that are not in the table yet
                unsigned need = S & ~have;
                while (need) {
                    const int blk = __builtin_ctz(need);
                    need &= need - 1;
                    int slot = -1;
#pragma unroll
                    for (int sl = 0; sl < NL; ++sl)
                        if (slot < 0 && tab[sl] == EMPTY) slot = sl;
                    if (slot < 0) {
#pragma unroll
                        for (int sl = 0; sl < NL; ++sl)
                            if (slot < 0 && !((S >> tab[sl]) & 1u)) slot = sl;
                        flush_dyn(slot);
                    }
                    set_slot(slot, blk);
                    have |= 1u << blk;
                }
                unsigned L = 0;
#pragma unroll
                for (int sl = 0; sl < NL; ++sl)
                    if (tab[sl] != EMPTY && ((S >> tab[sl]) & 1u)) L |= 1u << sl;
                process(ks0, L, L, pairs_within(L));
            } else {
                // dense spot: rounds over pairs of 2-block chunks, table rebuilt and flushed every round
                flush_all();
                unsigned cmw = 0, rem = S;  // four 8-bit masks of up to two blocks each
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    unsigned m2 = 0;
                    if (rem) m2 |= rem & (0u - rem), rem &= rem - 1;
                    if (rem) m2 |= rem & (0u - rem), rem &= rem - 1;
                    cmw |= m2 << (8 * a);
                }
#pragma unroll 1
                for (int a = 0; a < 4; ++a) {
                    const unsigned ma = (cmw >> (8 * a)) & 0xffu;
                    if (!ma) break;
#pragma unroll 1
                    for (int bq = a; bq < 4; ++bq) {
                        const unsigned mb = (cmw >> (8 * bq)) & 0xffu;
                        if (!mb) break;
                        const unsigned ma2 = ma & (ma - 1), mb2 = mb & (mb - 1);
                        tab[0] = __builtin_ctz(ma);
                        tab[1] = ma2 ? __builtin_ctz(ma2) : EMPTY;
                        tab[2] = bq != a ? __builtin_ctz(mb) : EMPTY;
                        tab[3] = (bq != a && mb2) ? __builtin_ctz(mb2) : EMPTY;
                        unsigned Lg = 0;
#pragma unroll
                        for (int sl = 0; sl < NL; ++sl)
                            if (tab[sl] != EMPTY) Lg |= 1u << sl, have |= 1u << tab[sl];
                        unsigned tm;
                        if (bq == a) {
                            tm = pairs_within(Lg & 3u);
                        } else {  // cross tiles only: (0,2) (0,3) (1,2) (1,3)
                            tm = 0;
                            if ((Lg & 5u) == 5u) tm |= 1u << lt(0, 2);
                            if ((Lg & 9u) == 9u) tm |= 1u << lt(0, 3);
                            if ((Lg & 6u) == 6u) tm |= 1u << lt(1, 2);
                            if ((Lg & 10u) == 10u) tm |= 1u << lt(1, 3);
                        }
                        process(ks0, Lg, bq == a ? (Lg & 3u) : 0u, tm);
                        flush_all();
                    }
                }
            }
        }
        if (++pz == p.npz) {
            pz = 0;
            if (++py == p.npy) py = 0, ++px;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    flush_all();
    if (p.counters && lane == 0) {
        atomicAdd(p.counters + 0, (unsigned long long)n_tiles * KP);
        atomicAdd(p.counters + 1, (unsigned long long)n_blocks * KP);
    }
}

// finish for the full-grid slab of warp_gram_lt_kernel: tile (bi,bj) plus the transposed tile (bj,bi)
template <int NB>
__global__ __launch_bounds__(256) void gram_lt_finish_kernel(const float *__restrict__ slab, int nchunks, int K,
                                                             const int *__restrict__ order, float *__restrict__ G,
                                                             float *__restrict__ r) {
    constexpr int SLAB = NB * NB * 256 + 128;
    const int b = blockIdx.x;
    const int e = threadIdx.x;
    const int lane = e >> 2, reg = e & 3;
    const int i = 4 * (lane >> 4) + reg, j = lane & 15;
    const int et = (((j >> 2) * 16 + i) << 2) + (j & 3);  // the same element in the transposed tile's layout
    float *Gb = G + (long)b * K * K;
    const float *base = slab + (long)b * nchunks * SLAB;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
            float s = 0.0f;
            for (int c = 0; c < nchunks; ++c) {
                s += base[(long)c * SLAB + (bi * NB + bj) * 256 + e];
                if (bi != bj) s += base[(long)c * SLAB + (bj * NB + bi) * 256 + et];
            }
            const int ck = 16 * bi + i, cl = 16 * bj + j;
            if (ck < K && cl < K) {
                const int k = order[ck], l = order[cl];
                Gb[(long)k * K + l] = s;
                if (bi != bj) Gb[(long)l * K + k] = s;
            }
        }
    }
    if (e < 16 * NB && e < K) {
        float s = 0.0f;
        for (int c = 0; c < nchunks; ++c) s += base[(long)c * SLAB + NB * NB * 256 + e];
        r[(long)b * K + order[e]] = s;
    }
}

// ordered chunk sum + scatter through the neuron order (sorted channel c -> neuron order[c])
template <int NB>
__global__ __launch_bounds__(256) void gram_sparse_finish_kernel(const float *__restrict__ slab, int nchunks, int K,
                                                                 const int *__restrict__ order, float *__restrict__ G,
                                                                 float *__restrict__ r) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int SLAB = NT * 256 + 128;
    const int b = blockIdx.x;
    const int e = threadIdx.x;
    const int lane = e >> 2, reg = e & 3;
    const int i = 4 * (lane >> 4) + reg, j = lane & 15;
    float *Gb = G + (long)b * K * K;
    const float *base = slab + (long)b * nchunks * SLAB;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
            const float *src = base + sp_tile_index(NB, bi, bj) * 256 + e;
            float s = 0.0f;
            for (int c = 0; c < nchunks; ++c) s += src[(long)c * SLAB];
            const int ck = 16 * bi + i, cl = 16 * bj + j;
            if (ck < K && cl < K) {
                const int k = order[ck], l = order[cl];
                Gb[(long)k * K + l] = s;
                Gb[(long)l * K + k] = s;
            }
        }
    }
    if (e < 16 * NB && e < K) {
        float s = 0.0f;
        for (int c = 0; c < nchunks; ++c) s += base[(long)c * SLAB + NT * 256 + e];
        r[(long)b * K + order[e]] = s;
    }
}

static void sp_patch_shape(const Volume &vol, int &lgy, int &lgz, int &npy, int &npz, long &npatch) {
    lgz = vol.Z == 1 ? 0 : (vol.Z == 2 ? 1 : 2);
    lgy = vol.Z <= 2 ? 3 : 2;
    const int lgx = 6 - lgy - lgz;
    npz = (vol.Z + (1 << lgz) - 1) >> lgz;
    npy = (vol.Y + (1 << lgy) - 1) >> lgy;
    npatch = (long)((vol.X + (1 << lgx) - 1) >> lgx) * npy * npz;
}

static void sp_choose_chunks(long npatch, int B, int &nchunks, long &chunk_len) {
    const long nss = npatch;
    long want = (8192 + B - 1) / B;  // passes differ a lot in cost: more, smaller work items than the dense kernel
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    if (want > nss) want = nss;
    const long ss_per_chunk = (nss + want - 1) / want;
    chunk_len = ss_per_chunk;
    nchunks = (int)((npatch + chunk_len - 1) / chunk_len);
}

template <int NB>
static int launch_sparse(SparseParams p, const int *order, float *G, float *r, hipStream_t st) {
    const long nitems = (long)p.nchunks * p.B;
    const unsigned nwg = (unsigned)((nitems + 3) / 4);
    if (p.vol.Z > 1)
        hipLaunchKernelGGL((warp_gram_sparse_kernel<NB, 8>), dim3(nwg), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((warp_gram_sparse_kernel<NB, 4>), dim3(nwg), dim3(256), 0, st, p);
    hipLaunchKernelGGL((gram_sparse_finish_kernel<NB>), dim3((unsigned)p.B), dim3(256), 0, st, p.slab, p.nchunks, p.K,
                       order, G, r);
    return check_launch("dnmf_warp_gram_rhs_sparse");
}

template <int NB>
static int launch_sparse_lt(SparseParams p, const int *order, float *G, float *r, hipStream_t st) {
    const long nitems = (long)p.nchunks * p.B;
    const unsigned nwg = (unsigned)((nitems + 3) / 4);
    const size_t bytes = (size_t)nitems * ((size_t)NB * NB * 256 + 128) * sizeof(float);
    hipError_t e = hipMemsetAsync(p.slab, 0, bytes, st);  // the kernel adds into its slab region
    if (e != hipSuccess) return fail((int)e, "dnmf_warp_gram_rhs_sparse_lt: memset: %s", hipGetErrorString(e));
    if (p.vol.Z > 1)
        hipLaunchKernelGGL((warp_gram_lt_kernel<NB, 8>), dim3(nwg), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((warp_gram_lt_kernel<NB, 4>), dim3(nwg), dim3(256), 0, st, p);
    hipLaunchKernelGGL((gram_lt_finish_kernel<NB>), dim3((unsigned)p.B), dim3(256), 0, st, p.slab, p.nchunks, p.K, order, G,
                       r);
    return check_launch("dnmf_warp_gram_rhs_sparse_lt");
}

}  // namespace dnmf

extern "C" {

size_t dnmf_warp_gram_rhs_sparse_lt_workspace(long P, int K, int B) {
    if (P <= 0 || K <= 0 || B <= 0) return 0;
    const int NB = (K + 15) / 16;
    long want = (8192 + B - 1) / B;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return (size_t)B * (size_t)(want + 1) * ((size_t)NB * NB * 256 + 128) * sizeof(float);
}

int dnmf_warp_gram_rhs_sparse_lt(const float *Aps, int Ks, int K, const int *order, const unsigned char *row_mask, int X,
                                 int Y, int Z, const float *beta, int T, const int *times, int B, const float *frames,
                                 long ldf, const int *frame_ids, float *G, float *r, void *workspace,
                                 size_t workspace_bytes, unsigned long long *counters, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Aps && order && row_mask && beta && frames && G && r && workspace, DNMF_E_NULL,
                 "dnmf_warp_gram_rhs_sparse_lt: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && Ks == 16 * ((K + 15) / 16), DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_sparse_lt: X=%d Y=%d Z=%d K=%d Ks=%d T=%d B=%d", X, Y, Z, K, Ks, T, B);
    DNMF_REQUIRE(Ks <= 128, DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_sparse_lt: K=%d > 128", K);
    SparseParams p;
    p.vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= p.vol.P, DNMF_E_SHAPE, "dnmf_warp_gram_rhs_sparse_lt: ldf=%ld < P=%ld", ldf, p.vol.P);
    DNMF_REQUIRE(p.vol.P * Ks < (1L << 30), DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_sparse_lt: P*Ks=%ld too large",
                 p.vol.P * Ks);
    DNMF_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_sparse_lt: workspace must be 16-byte aligned");
    p.Aps = Aps, p.row_mask = row_mask, p.Ks = Ks, p.K = K;
    p.beta = beta, p.T = T, p.times = times, p.B = B;
    p.frames = frames, p.ldf = ldf, p.frame_ids = frame_ids;
    p.slab = static_cast<float *>(workspace);
    p.counters = counters;
    sp_patch_shape(p.vol, p.lgy, p.lgz, p.npy, p.npz, p.npatch);
    sp_choose_chunks(p.npatch, B, p.nchunks, p.chunk_len);
    const int NB = Ks / 16;
    DNMF_REQUIRE(workspace_bytes >= (size_t)B * p.nchunks * ((size_t)NB * NB * 256 + 128) * sizeof(float), DNMF_E_WORKSPACE,
                 "dnmf_warp_gram_rhs_sparse_lt: workspace too small for %d chunks", p.nchunks);
    hipStream_t st = (hipStream_t)stream;
    switch (NB) {
        case 1: return launch_sparse_lt<1>(p, order, G, r, st);
        case 2: return launch_sparse_lt<2>(p, order, G, r, st);
        case 3: return launch_sparse_lt<3>(p, order, G, r, st);
        case 4: return launch_sparse_lt<4>(p, order, G, r, st);
        case 5: return launch_sparse_lt<5>(p, order, G, r, st);
        case 6: return launch_sparse_lt<6>(p, order, G, r, st);
        case 7: return launch_sparse_lt<7>(p, order, G, r, st);
        default: return launch_sparse_lt<8>(p, order, G, r, st);
    }
}

int dnmf_sparse_k(int K) { return K < 1 ? 0 : 16 * ((K + 15) / 16); }

int dnmf_pack_footprints_sparse(const float *A, long P, int K, const int *order, float *Aps, int Ks,
                                unsigned char *row_mask, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && order && Aps && row_mask, DNMF_E_NULL, "dnmf_pack_footprints_sparse: NULL buffer");
    DNMF_REQUIRE(P > 0 && K > 0 && Ks == dnmf_sparse_k(K), DNMF_E_SHAPE, "dnmf_pack_footprints_sparse: P=%ld K=%d Ks=%d", P,
                 K, Ks);
    DNMF_REQUIRE(Ks <= 128, DNMF_E_UNSUPPORTED, "dnmf_pack_footprints_sparse: K=%d > 128 (8 blocks per mask byte)", K);
    const long nthreads = P * 16;
    hipLaunchKernelGGL(pack_sparse_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A,
                       P, K, order, Aps, Ks, row_mask);
    return check_launch("dnmf_pack_footprints_sparse");
}

size_t dnmf_warp_gram_rhs_sparse_workspace(long P, int K, int B) {
    if (P <= 0 || K <= 0 || B <= 0) return 0;
    const int NB = dnmf_sparse_k(K) / 16;
    // the patch count depends on the volume shape, not only on P: bound it by the worst shape (all three
    // axes one voxel past a patch boundary cannot exceed 8x the voxel count / 64)
    const long npatch_max = P / 8 + 64;
    const long per_item = (long)(NB * (NB + 1) / 2) * 256 + 128;
    long want = (8192 + B - 1) / B;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    (void)npatch_max;
    return (size_t)B * (size_t)(want + 1) * per_item * sizeof(float);
}

int dnmf_warp_gram_rhs_sparse(const float *Aps, int Ks, int K, const int *order, const unsigned char *row_mask, int X,
                              int Y, int Z, const float *beta, int T, const int *times, int B, const float *frames,
                              long ldf, const int *frame_ids, float *G, float *r, void *workspace,
                              size_t workspace_bytes, unsigned long long *counters, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Aps && order && row_mask && beta && frames && G && r && workspace, DNMF_E_NULL,
                 "dnmf_warp_gram_rhs_sparse: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && Ks == dnmf_sparse_k(K), DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_sparse: X=%d Y=%d Z=%d K=%d Ks=%d T=%d B=%d", X, Y, Z, K, Ks, T, B);
    DNMF_REQUIRE(Ks <= 128, DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_sparse: K=%d > 128", K);
    SparseParams p;
    p.vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= p.vol.P, DNMF_E_SHAPE, "dnmf_warp_gram_rhs_sparse: ldf=%ld < P=%ld", ldf, p.vol.P);
    DNMF_REQUIRE(p.vol.P * Ks < (1L << 30), DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_sparse: P*Ks=%ld too large",
                 p.vol.P * Ks);
    DNMF_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_sparse: workspace must be 16-byte aligned");
    DNMF_REQUIRE(workspace_bytes >= dnmf_warp_gram_rhs_sparse_workspace(p.vol.P, K, B), DNMF_E_WORKSPACE,
                 "dnmf_warp_gram_rhs_sparse: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_warp_gram_rhs_sparse_workspace(p.vol.P, K, B));
    p.Aps = Aps, p.row_mask = row_mask, p.Ks = Ks, p.K = K;
    p.beta = beta, p.T = T, p.times = times, p.B = B;
    p.frames = frames, p.ldf = ldf, p.frame_ids = frame_ids;
    p.slab = static_cast<float *>(workspace);
    p.counters = counters;
    sp_patch_shape(p.vol, p.lgy, p.lgz, p.npy, p.npz, p.npatch);
    sp_choose_chunks(p.npatch, B, p.nchunks, p.chunk_len);
    DNMF_REQUIRE(workspace_bytes >= (size_t)B * p.nchunks * ((size_t)(Ks / 16 * (Ks / 16 + 1) / 2) * 256 + 128) * sizeof(float),
                 DNMF_E_WORKSPACE, "dnmf_warp_gram_rhs_sparse: workspace too small for %d chunks", p.nchunks);
    hipStream_t st = (hipStream_t)stream;
    switch (Ks / 16) {
        case 1: return launch_sparse<1>(p, order, G, r, st);
        case 2: return launch_sparse<2>(p, order, G, r, st);
        case 3: return launch_sparse<3>(p, order, G, r, st);
        case 4: return launch_sparse<4>(p, order, G, r, st);
        case 5: return launch_sparse<5>(p, order, G, r, st);
        case 6: return launch_sparse<6>(p, order, G, r, st);
        case 7: return launch_sparse<7>(p, order, G, r, st);
        default: return launch_sparse<8>(p, order, G, r, st);
    }
}

}  // extern "C"
