#!/usr/bin/env python3
"""Regenerate the timing-only ablation sources used by tools/run_ablation.sh and tools/run_ablation_sparse.sh from the
current kernels (the results of these builds are wrong on purpose; only their run time is read):
  tools/abl_warp_gram_rhs.hip     dense Gram kernel; -DABL=1 drops the blend of taps 1-3 (and with it their loads),
                                  -DABL=2 loads only tap 0 but keeps the blend
  tools/abl_warp_gram_sparse.hip  table Gram kernel with its gathers served from LDS instead of global memory
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dnmf_amd", "csrc")


def must_replace(s, old, new):
    assert old in s, old[:60]
    return s.replace(old, new)


def main():
    s = open(os.path.join(CSRC, "warp_gram_rhs.hip")).read()
    s = must_replace(s, '''#pragma unroll
        for (int c = 1; c < NTAP; ++c) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) frag[bb] = fmaf(st.raw[c].v[bb], st.w[c >> 2][c & 3], frag[bb]);
        }''', '''#if ABL != 1
#pragma unroll
        for (int c = 1; c < NTAP; ++c) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) frag[bb] = fmaf(st.raw[c].v[bb], st.w[c >> 2][c & 3], frag[bb]);
        }
#endif''')
    s = must_replace(s, '''            for (int e = 0; e < 4; ++e) load_row<NB>(st.raw[4 * q + e], Ab, rc.rr[q][e] + lane_a, rc.rr[q][e] + lane_b);''',
                     '''#if ABL == 2
            load_row<NB>(st.raw[4 * q], Ab, rc.rr[q][0] + lane_a, rc.rr[q][0] + lane_b);
            for (int e = 1; e < 4; ++e) st.raw[4 * q + e] = st.raw[4 * q];
#else
            for (int e = 0; e < 4; ++e) load_row<NB>(st.raw[4 * q + e], Ab, rc.rr[q][e] + lane_a, rc.rr[q][e] + lane_b);
#endif''')
    open(os.path.join(ROOT, "tools", "abl_warp_gram_rhs.hip"), "w").write(
        s.replace('#include "common.hpp"', '#include "../dnmf_amd/csrc/common.hpp"'))

    s = open(os.path.join(CSRC, "warp_gram_sparse.hip")).read()
    i = s.index('    auto process = [&](int ks0, unsigned Lg, unsigned Lr, unsigned tm) {')
    j = s.index('    auto pairs_within = [&](unsigned L) {')
    blk = must_replace(s[i:j], 'v = fmaf(*reinterpret_cast<const float *>(Ab + (size_t)(rr[e] + boff)), ww[e], v);',
                       'v = fmaf(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(&s_w[wave][0][0]) + '
                       '((rr[e] + boff) & 1020u)), ww[e], v);')
    s = s[:i] + blk + s[j:]
    open(os.path.join(ROOT, "tools", "abl_warp_gram_sparse.hip"), "w").write(
        s.replace('#include "common.hpp"', '#include "../dnmf_amd/csrc/common.hpp"'))


if __name__ == "__main__":
    main()
