// K3n -- fused warp + per-frame Gram matrix + right-hand side for COMPACT footprints, neuron by neuron.
//
// Reference: the two contractions of DeformableNMF.update_temporal (Demix/dNMF.py:141-142) on the warped
// footprints of spatial_pushforward (dNMF.py:69-87), as K3 / K3s.
//
// Why another kernel.  The reference's Gaussians underflow to an exact zero ~30 px from their centre, so at
// 512x512, K=100 a voxel sees 1.1 non-zero footprints on average and A_t^T A_t has ~2.6 GFLOP of non-zero products
// per 4000 frames, not 11.6 TFLOP.  K3s skips whole 16-neuron blocks but still evaluates 16x16 products per active
// block pair on the matrix pipe and spends most of its time on block bookkeeping.  Here nothing is padded to
// blocks: a wave walks over tiles of 256 voxels, works out which neurons can reach the tile at all and evaluates
// only those, on the vector ALU -- no MFMA: there is no dense operand left to feed it.
//
// Exactness.  A product is skipped only when one factor is an exact zero, so every sum equals the dense kernel's
// up to the order of fp32 additions:
//   - bbox[k] is the bounding box of the non-zeros of footprint k (dnmf_pack_footprints_lists);
//   - the taps of a tile's voxels are computed with the reference's fp32 coordinate sequence (common.hpp) FIRST,
//     and the tile's neuron list is every k whose bbox meets the box [min tap, max tap] of those actual taps;
//   - G[k,l] can be non-zero for some warp only if a 2x2(x2) tap cell meets both bboxes, i.e. if per axis
//     k_lo - 1 <= l_hi and l_lo - 1 <= k_hi: a static pattern, independent of the warp (both footprints move
//     under the same map).  Pairs outside the pattern are never accumulated and come out as 0.
//
// Data flow.  Lane = voxel (16 lanes along y: the neuron-major footprint copy At (K,P) is read in coalesced 64-byte
// runs), four voxels per lane.  Per listed neuron: 4 x NTAP loads and FMAs give its warped values a_k at the lane's
// voxels, then r_k += a_k.y and G_kl += a_k.a_l for the listed l >= k: per-lane partial sums, a fixed DPP tree over
// the 64 lanes, and one LDS add by the last lane into the wave's private table of pattern slots.  LDS operations of
// one wave retire in order, so the sum is deterministic.  The table goes to a slab per (frame, chunk); the finish
// kernel adds the chunks in order and scatters the slots into dense G (K,K), r (K).
#include <type_traits>

#include "common.hpp"

namespace dnmf {

constexpr int LISTS_NG = 4;      // neurons evaluated together (register slots); longer lists are cut into groups
constexpr int LISTS_LGV = 2;      // log2 of the voxels per lane (consecutive x positions, interleaved by lane)
constexpr int LISTS_VPL = 1 << LISTS_LGV;
constexpr int LISTS_MAXW = 4;    // 64-neuron words of a tile's list: K <= 256
constexpr long LISTS_ITEMS = 16384;  // target number of wave-sized work items per launch
constexpr int LISTS_MAX_SLOTS = 3800;  // 4 waves x (3800 + 256) words of LDS per workgroup

struct ListParams {
    const float *At;       // (K,P)
    const int *bbox;       // (K,6) xlo,xhi,ylo,yhi,zlo,zhi; lo > hi for an all-zero footprint
    const int *pair_slot;  // (K,K) symmetric; pairs outside the pattern point at the trash slot nslot-1
    int nslot;             // K rhs slots, then the pattern pairs, then one trash slot
    int K;
    Volume vol;
    const float *beta;
    int T;
    const int *times;
    int B;
    const float *frames;
    long ldf;
    const int *frame_ids;
    float *slab;  // (B, nchunks, nslot)
    int nchunks, chunk_len;
    int lgx, lgz;  // tile = (LISTS_VPL << lgx) x 16 x (1 << lgz) voxels, lgx + lgz = 2
    int nty, ntz, ntiles;
    unsigned long long *counters;  // optional: [0] += (tile, neuron) evaluations, [1] += (tile, pair) sums
};

// sum over the 64 lanes, valid in lane 63 (fixed tree: row prefix sums, then the row totals)
__device__ __forceinline__ float wave_sum_last(float v) {
#define DNMF_STEP(ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true));
    DNMF_STEP(0x111, 0xf)  // row_shr:1
    DNMF_STEP(0x112, 0xf)  // row_shr:2
    DNMF_STEP(0x114, 0xf)  // row_shr:4
    DNMF_STEP(0x118, 0xf)  // row_shr:8 -> lane 15 of a row holds the row sum
    DNMF_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    DNMF_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef DNMF_STEP
    return v;
}

// min / max over the 64 lanes, valid in lane 63
template <bool MAX>
__device__ __forceinline__ int wave_minmax_last(int v) {
    constexpr int ident = MAX ? (int)0x80000000 : 0x7fffffff;
#define DNMF_STEP(ctrl, rmask)                                                          \
    {                                                                                   \
        const int o = __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false);   \
        v = MAX ? max(v, o) : min(v, o);                                                \
    }
    DNMF_STEP(0x111, 0xf)
    DNMF_STEP(0x112, 0xf)
    DNMF_STEP(0x114, 0xf)
    DNMF_STEP(0x118, 0xf)
    DNMF_STEP(0x142, 0xa)
    DNMF_STEP(0x143, 0xc)
#undef DNMF_STEP
    return v;
}

template <int NTAP, int NW, int FAST>
__global__ __launch_bounds__(256) void warp_gram_lists_kernel(ListParams p) {
    extern __shared__ float s_tab[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;  // chunk-major: neighbouring waves work on the same part of At
    if (item >= (long)p.nchunks * p.B) return;      // whole wave leaves; no workgroup barrier below
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const int K = p.K;
    const size_t plane = (size_t)vol.P * 4u;  // bytes of one neuron's footprint

    float bt[30];
    load_beta(p.beta, p.T, t, bt);

    float *tab = s_tab + (size_t)wave * p.nslot;
    for (int i = lane; i < p.nslot; i += 64) tab[i] = 0.0f;
    const unsigned tab_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tab;  // LDS byte address
    int *lst = reinterpret_cast<int *>(s_tab + (size_t)4 * p.nslot) + wave * (64 * NW);  // this tile's neuron list

    // the neurons this lane tests against a tile's tap box: k = lane + 64 w
    int bx[NW][6];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int k = lane + 64 * w;
#pragma unroll
        for (int e = 0; e < 6; ++e) bx[w][e] = k < K ? p.bbox[k * 6 + e] : ((e & 1) ? -1 : 0x7fffffff);
    }

    const int lgx = p.lgx, lgz = p.lgz;
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & 15, lx = lane >> (lgz + 4);
    const int q_begin = chunk * p.chunk_len;
    const int q_end = min(q_begin + p.chunk_len, p.ntiles);
    unsigned long long n_eval = 0, n_pair = 0;  // wave-uniform

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int q = q_begin; q < q_end; ++q) {
        const int qz = q % p.ntz, qy = (q / p.ntz) % p.nty, qx = q / (p.ntz * p.nty);
        const int y = (qy << 4) + ly, z = (qz << lgz) + lz;

        // ---- taps of this lane's four voxels ------------------------------------------------------------
        unsigned off[LISTS_VPL][NTAP];  // byte offset of the tap inside a neuron's plane
        float w[LISTS_VPL][NTAP];
        float yv[LISTS_VPL];
        int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
        for (int v = 0; v < LISTS_VPL; ++v) {
            const int x = (qx << (lgx + LISTS_LGV)) + (v << lgx) + lx;
#pragma unroll
            for (int c = 0; c < NTAP; ++c) off[v][c] = 0u, w[v][c] = 0.0f;
            yv[v] = 0.0f;
            if (x < vol.X && y < vol.Y && z < vol.Z) {
                const Sample sm = make_sample_t<(NTAP == 8), FAST>(bt, vol, x, y, z);
                unsigned vox[NTAP];
                make_taps<NTAP>(sm, vol, w[v], vox);
#pragma unroll
                for (int c = 0; c < NTAP; ++c) off[v][c] = vox[c] * 4u;
                yv[v] = yb[((long)x * vol.Y + y) * vol.Z + z];
                mn[0] = min(mn[0], sm.x0), mx[0] = max(mx[0], sm.x0);
                mn[1] = min(mn[1], sm.y0), mx[1] = max(mx[1], sm.y0);
                if (NTAP == 8) mn[2] = min(mn[2], sm.z0), mx[2] = max(mx[2], sm.z0);
            }
        }
        // box of the (clamped) tap coordinates of the whole tile; an all-invalid lane contributes the identities
        int lo[3], hi[3];
#pragma unroll
        for (int d = 0; d < (NTAP == 8 ? 3 : 2); ++d) {
            const int size = d == 0 ? vol.X : (d == 1 ? vol.Y : vol.Z);
            const int a = __builtin_amdgcn_readlane(wave_minmax_last<false>(mn[d]), 63);
            const int c = __builtin_amdgcn_readlane(wave_minmax_last<true>(mx[d]), 63);
            lo[d] = min(max(a, 0), size - 1);
            hi[d] = min(max(c + 1, 0), size - 1);
        }
        if (NTAP != 8) lo[2] = 0, hi[2] = 0;

        // ---- the tile's neuron list: ascending neuron indices -------------------------------------------------
        // Up to LISTS_NG neurons (nearly every tile) are picked out of the ballot masks with scalar bit operations;
        // longer lists are compacted into the wave's LDS strip.
        unsigned long long msk[NW];
        int n = 0;  // wave-uniform
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) {
            const bool hit = bx[wd][0] <= hi[0] && bx[wd][1] >= lo[0] && bx[wd][2] <= hi[1] && bx[wd][3] >= lo[1] &&
                             bx[wd][4] <= hi[2] && bx[wd][5] >= lo[2];
            msk[wd] = __ballot(hit);
            n += __builtin_popcountll(msk[wd]);
        }
        if (n == 0) continue;
        if (n > LISTS_NG) {
            int at = 0;
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) {
                const unsigned long long m = msk[wd];
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if ((m >> lane) & 1) lst[at + before] = lane + 64 * wd;
                at += __builtin_popcountll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // neuron ids of the group that starts at list position g (wave-uniform scalars; -1 past the end)
        auto group_ids = [&](int g, int (&ks)[LISTS_NG]) {
            if (n <= LISTS_NG) {  // g == 0: lowest set bits of the masks, in order
                unsigned long long rem[NW];
#pragma unroll
                for (int wd = 0; wd < NW; ++wd) rem[wd] = msk[wd];
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) {
                    int k = -1;
                    bool found = false;
#pragma unroll
                    for (int wd = 0; wd < NW; ++wd) {
                        const bool take = !found && rem[wd] != 0;
                        k = take ? 64 * wd + __builtin_ctzll(rem[wd]) : k;
                        rem[wd] = take ? rem[wd] & (rem[wd] - 1) : rem[wd];
                        found = found || take;
                    }
                    ks[i] = k;
                }
                return;
            }
            const int mine = lst[min(g + (lane & 7), n - 1)];
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i) ks[i] = g + i < n ? __builtin_amdgcn_readlane(mine, i) : -1;
        };
        auto eval = [&](int k, float (&a)[LISTS_VPL]) {
            const char *__restrict__ Ak = reinterpret_cast<const char *>(p.At) + (size_t)k * plane;
#pragma unroll
            for (int v = 0; v < LISTS_VPL; ++v) {
                // the offsets are re-materialised as 32-bit values here so that the loads take the
                // (scalar base + 32-bit vector offset) form; hoisted out of the neuron loop they become 64-bit pairs
                unsigned o[NTAP];
#pragma unroll
                for (int c = 0; c < NTAP; ++c) {
                    o[c] = off[v][c];
                    asm("" : "+v"(o[c]));
                }
                float s = *reinterpret_cast<const float *>(Ak + o[0]) * w[v][0];
#pragma unroll
                for (int c = 1; c < NTAP; ++c) s = fmaf(*reinterpret_cast<const float *>(Ak + o[c]), w[v][c], s);
                a[v] = s;
            }
        };
        auto dot4 = [&](const float (&a)[LISTS_VPL], const float (&c)[LISTS_VPL]) {
            float s = a[0] * c[0];
#pragma unroll
            for (int v = 1; v < LISTS_VPL; ++v) s = fmaf(a[v], c[v], s);
            return s;
        };
        // one lane, one LDS add (an atomic builtin here is rewritten into a cross-lane reduction loop)
        auto add_slot = [&](int slot, float total_in_last_lane) {
            const unsigned addr = tab_lds + 4u * (unsigned)slot;
            if (lane == 63) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(total_in_last_lane) : "memory");
        };
        // slot of a pair: a scalar load issued before the arithmetic that precedes its use
        auto pair_slot_of = [&](int k, int l) {
            return __builtin_amdgcn_readfirstlane(p.pair_slot[max(k, 0) * K + max(l, 0)]);
        };
        // the N leading neurons of a group: their warped values, then the sums among themselves and against the frame;
        // straight-line code, so that the row requests overlap and the N + N(N+1)/2 reduction trees interleave
        auto within = [&](auto nn, const int (&ks)[LISTS_NG], float (&a)[LISTS_NG][LISTS_VPL]) {
            constexpr int N = decltype(nn)::value;
#pragma unroll
            for (int i = 0; i < N; ++i) eval(ks[i], a[i]);  // the rows of all N neurons are requested together
            int sl[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = i; j < N; ++j) sl[i][j] = pair_slot_of(ks[i], ks[j]);
            float sr[N], sp[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                sr[i] = wave_sum_last(dot4(a[i], yv));
#pragma unroll
                for (int j = i; j < N; ++j) sp[i][j] = wave_sum_last(dot4(a[i], a[j]));
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                add_slot(ks[i], sr[i]);
#pragma unroll
                for (int j = i; j < N; ++j) add_slot(sl[i][j], sp[i][j]);
            }
        };

        n_eval += n, n_pair += n * (n + 1) / 2;
        for (int g1 = 0; g1 < n; g1 += LISTS_NG) {
            int kA[LISTS_NG];
            float aA[LISTS_NG][LISTS_VPL];
            group_ids(g1, kA);
            using std::integral_constant;
            switch (min(n - g1, LISTS_NG)) {
                case 1: within(integral_constant<int, 1>{}, kA, aA); break;
                case 2: within(integral_constant<int, 2>{}, kA, aA); break;
                case 3: within(integral_constant<int, 3>{}, kA, aA); break;
                default: within(integral_constant<int, 4>{}, kA, aA); break;
            }
            static_assert(LISTS_NG == 4, "the dispatch above lists the group sizes");
            // pairs of this (then full) group with every later group
            for (int g2 = g1 + LISTS_NG; g2 < n; g2 += LISTS_NG) {
                int kB[LISTS_NG];
                float aB[LISTS_NG][LISTS_VPL];
                group_ids(g2, kB);
#pragma unroll
                for (int j = 0; j < LISTS_NG; ++j) {
                    if (kB[j] < 0) continue;
                    eval(kB[j], aB[j]);
                    int sl[LISTS_NG];
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i) sl[i] = pair_slot_of(kA[i], kB[j]);
                    float sp[LISTS_NG];
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i) sp[i] = wave_sum_last(dot4(aA[i], aB[j]));
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i) add_slot(sl[i], sp[i]);
                }
            }
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float *out = p.slab + ((long)b * p.nchunks + chunk) * p.nslot;
    for (int i = lane; i < p.nslot; i += 64) out[i] = tab[i];
    if (p.counters && lane == 0) {
        atomicAdd(&p.counters[0], n_eval);
        atomicAdd(&p.counters[1], n_pair);
    }
}

// G[b] (K,K), r[b] (K) <- ordered sum of the chunk tables of frame b
__global__ __launch_bounds__(256) void gram_lists_finish_kernel(const float *__restrict__ slab, int nchunks, int nslot,
                                                                const int *__restrict__ pair_slot, int K,
                                                                float *__restrict__ G, float *__restrict__ r) {
    const int b = blockIdx.x;
    const float *src = slab + (long)b * nchunks * nslot;
    for (int e = threadIdx.x; e < K * K + K; e += blockDim.x) {
        const int slot = e < K * K ? pair_slot[e] : e - K * K;
        float s = 0.0f;
        if (slot != nslot - 1)
            for (int c = 0; c < nchunks; ++c) s += src[(long)c * nslot + slot];
        if (e < K * K)
            G[(long)b * K * K + e] = s;
        else
            r[(long)b * K + (e - K * K)] = s;
    }
}

// ---- layout ----------------------------------------------------------------------------------------------
__global__ void lists_init_kernel(int *bbox, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * 6) bbox[i] = (i & 1) ? -1 : 0x7fffffff;
}

// At[k][p] = A[p][k] through a 32x32 LDS tile; bbox[k] grows over the non-zeros
__global__ __launch_bounds__(256) void lists_transpose_kernel(const float *__restrict__ A, long P, int K, Volume vol,
                                                              float *__restrict__ At, int *__restrict__ bbox) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const long pp = p0 + i;
        const int k = k0 + tx;
        tile[i][tx] = (pp < P && k < K) ? A[pp * K + k] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i;
        const long pp = p0 + tx;
        if (k < K && pp < P) {
            const float v = tile[tx][i];
            At[(long)k * P + pp] = v;
            if (v != 0.0f) {
                int x, y, z;
                voxel_xyz(pp, vol, x, y, z);
                atomicMin(&bbox[k * 6 + 0], x), atomicMax(&bbox[k * 6 + 1], x);
                atomicMin(&bbox[k * 6 + 2], y), atomicMax(&bbox[k * 6 + 3], y);
                atomicMin(&bbox[k * 6 + 4], z), atomicMax(&bbox[k * 6 + 5], z);
            }
        }
    }
}

// pattern of G and its slots: one block, thread k owns row k.  Slots: [0,K) rhs, then (k,k), then (k,l>k) in order.
__global__ __launch_bounds__(256) void lists_pairs_kernel(const int *__restrict__ bbox, int K, int *__restrict__ pair_slot,
                                                          int *__restrict__ nslot_out) {
    __shared__ int cnt[256];
    __shared__ int base[257];
    const int k = threadIdx.x;
    auto meets = [&](int a, int c) {
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int alo = bbox[a * 6 + 2 * d], ahi = bbox[a * 6 + 2 * d + 1];
            const int clo = bbox[c * 6 + 2 * d], chi = bbox[c * 6 + 2 * d + 1];
            ok = ok && alo <= ahi && clo <= chi && alo - 1 <= chi && clo - 1 <= ahi;
        }
        return ok;
    };
    int n = 0;
    if (k < K) {
        n = 1;  // (k,k)
        for (int l = k + 1; l < K; ++l) n += meets(k, l) ? 1 : 0;
    }
    cnt[k] = n;
    __syncthreads();
    if (k == 0) {
        int s = K;
        for (int i = 0; i < 256; ++i) base[i] = s, s += cnt[i];
        base[256] = s;  // trash slot
        *nslot_out = s + 1;
    }
    __syncthreads();
    if (k < K) {
        const int trash = base[256];
        int s = base[k];
        pair_slot[k * K + k] = s++;
        for (int l = k + 1; l < K; ++l) {
            const int v = meets(k, l) ? s++ : trash;
            pair_slot[k * K + l] = v;
            pair_slot[l * K + k] = v;
        }
    }
}

static void lists_tile_shape(const Volume &vol, int &lgx, int &lgz, int &nty, int &ntz, int &ntiles) {
    lgz = vol.Z == 1 ? 0 : (vol.Z == 2 ? 1 : 2);
    lgx = 2 - lgz;
    const int tx = LISTS_VPL << lgx, tz = 1 << lgz;
    ntz = (vol.Z + tz - 1) / tz;
    nty = (vol.Y + 15) / 16;
    ntiles = ((vol.X + tx - 1) / tx) * nty * ntz;
}

static void lists_choose_chunks(int ntiles, int B, int &nchunks, int &chunk_len) {
    long want = (LISTS_ITEMS + B - 1) / B;  // wave-sized work items: several rounds of four waves per SIMD
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    if (want > ntiles) want = ntiles;
    chunk_len = (int)((ntiles + want - 1) / want);
    nchunks = (ntiles + chunk_len - 1) / chunk_len;
}

template <int NTAP, int NW>
static void launch_lists_t(const ListParams &p, unsigned nwg, size_t lds, hipStream_t st) {
    if (p.vol.fastdiv)
        hipLaunchKernelGGL((warp_gram_lists_kernel<NTAP, NW, 1>), dim3(nwg), dim3(256), lds, st, p);
    else
        hipLaunchKernelGGL((warp_gram_lists_kernel<NTAP, NW, 0>), dim3(nwg), dim3(256), lds, st, p);
}

}  // namespace dnmf

extern "C" {

int dnmf_pack_footprints_lists(const float *A, int X, int Y, int Z, int K, float *At, int *bbox, int *pair_slot,
                               int *nslot, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && At && bbox && pair_slot && nslot, DNMF_E_NULL, "dnmf_pack_footprints_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0, DNMF_E_SHAPE, "dnmf_pack_footprints_lists: X=%d Y=%d Z=%d K=%d", X, Y, Z,
                 K);
    DNMF_REQUIRE(K <= 64 * LISTS_MAXW, DNMF_E_UNSUPPORTED, "dnmf_pack_footprints_lists: K=%d > %d", K, 64 * LISTS_MAXW);
    const Volume vol = make_volume(X, Y, Z);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lists_init_kernel, dim3((K * 6 + 255) / 256), dim3(256), 0, st, bbox, K);
    hipLaunchKernelGGL(lists_transpose_kernel, dim3((unsigned)((vol.P + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0,
                       st, A, vol.P, K, vol, At, bbox);
    hipLaunchKernelGGL(lists_pairs_kernel, dim3(1), dim3(256), 0, st, bbox, K, pair_slot, nslot);
    return check_launch("dnmf_pack_footprints_lists");
}

size_t dnmf_warp_gram_rhs_lists_workspace(int nslot, int B) {
    if (nslot <= 0 || B <= 0) return 0;
    long want = (dnmf::LISTS_ITEMS + B - 1) / B;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return (size_t)B * (size_t)want * (size_t)nslot * sizeof(float);
}

int dnmf_warp_gram_rhs_lists(const float *At, const int *bbox, const int *pair_slot, int nslot, int K, int X, int Y, int Z,
                             const float *beta, int T, const int *times, int B, const float *frames, long ldf,
                             const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                             unsigned long long *counters, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(At && bbox && pair_slot && beta && frames && workspace && (!G == !r), DNMF_E_NULL,
                 "dnmf_warp_gram_rhs_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && nslot > K, DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_lists: X=%d Y=%d Z=%d K=%d T=%d B=%d nslot=%d", X, Y, Z, K, T, B, nslot);
    DNMF_REQUIRE(K <= 64 * LISTS_MAXW, DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_lists: K=%d > %d", K, 64 * LISTS_MAXW);
    DNMF_REQUIRE(nslot <= LISTS_MAX_SLOTS, DNMF_E_UNSUPPORTED,
                 "dnmf_warp_gram_rhs_lists: %d pattern slots > %d (footprints overlap too much for this kernel)", nslot,
                 LISTS_MAX_SLOTS);
    ListParams p;
    p.vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= p.vol.P, DNMF_E_SHAPE, "dnmf_warp_gram_rhs_lists: ldf=%ld < P=%ld", ldf, p.vol.P);
    DNMF_REQUIRE(p.vol.P < (1L << 30), DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_lists: P=%ld does not fit 32-bit offsets",
                 p.vol.P);
    p.At = At, p.bbox = bbox, p.pair_slot = pair_slot, p.nslot = nslot, p.K = K;
    p.beta = beta, p.T = T, p.times = times, p.B = B;
    p.frames = frames, p.ldf = ldf, p.frame_ids = frame_ids;
    p.slab = static_cast<float *>(workspace);
    p.counters = counters;
    lists_tile_shape(p.vol, p.lgx, p.lgz, p.nty, p.ntz, p.ntiles);
    lists_choose_chunks(p.ntiles, B, p.nchunks, p.chunk_len);
    DNMF_REQUIRE(workspace_bytes >= (size_t)B * p.nchunks * nslot * sizeof(float), DNMF_E_WORKSPACE,
                 "dnmf_warp_gram_rhs_lists: workspace %zu < %zu bytes", workspace_bytes,
                 (size_t)B * p.nchunks * nslot * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const long nitems = (long)p.nchunks * B;
    const unsigned nwg = (unsigned)((nitems + 3) / 4);
    const int nw = K <= 64 ? 1 : (K <= 128 ? 2 : 4);
    const size_t lds = (size_t)4 * (nslot + 64 * nw) * sizeof(float);
    if (Z > 1) {
        if (nw == 1) launch_lists_t<8, 1>(p, nwg, lds, st);
        else if (nw == 2) launch_lists_t<8, 2>(p, nwg, lds, st);
        else launch_lists_t<8, 4>(p, nwg, lds, st);
    } else {
        if (nw == 1) launch_lists_t<4, 1>(p, nwg, lds, st);
        else if (nw == 2) launch_lists_t<4, 2>(p, nwg, lds, st);
        else launch_lists_t<4, 4>(p, nwg, lds, st);
    }
    if (G)
        hipLaunchKernelGGL(gram_lists_finish_kernel, dim3((unsigned)B), dim3(256), 0, st, p.slab, p.nchunks, nslot,
                           pair_slot, K, G, r);
    return check_launch("dnmf_warp_gram_rhs_lists");
}

int dnmf_warp_gram_rhs_lists_chunks(int X, int Y, int Z, int B) {
    using namespace dnmf;
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    const Volume vol = make_volume(X, Y, Z);
    int lgx, lgz, nty, ntz, ntiles, nchunks, chunk_len;
    lists_tile_shape(vol, lgx, lgz, nty, ntz, ntiles);
    lists_choose_chunks(ntiles, B, nchunks, chunk_len);
    return nchunks;
}

}  // extern "C"
