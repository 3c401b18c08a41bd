#!/usr/bin/env python3
"""Writes a copy of rvdd-release_amd/csrc/conv3x3h.hip with the timing-only switches of the round-4 "what bounds a tile"
experiment (profiles/r04_conv3x3h_tile_loop_parts.txt): -DRVDD_XP=<bits>, 1 no MFMAs, 2 no halo loads / split / staging
stores in the loop, 4 no result stores, 8 fragment reads for the first two chunks only, 16 no epilogue arithmetic,
32 no barriers.  The shipping kernel carries none of them.
    python tools/conv3x3h_xp_patch.py tools/scratch/conv3x3h_xp.hip
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRVDD_STAMPS -DRVDD_XP=3 -DCONV_SRC='"scratch/conv3x3h_xp.hip"' \
          -Irvdd-release_amd/csrc -Itools tools/conv3x3h_bench.hip -o /tmp/c3hb && /tmp/c3hb 4 720 1280 1
"""
import os, sys

src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rvdd-release_amd", "csrc", "conv3x3h.hip")
s = open(src).read()


def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a)
    s = s.replace(a, b)


rep('''                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[cb][mt][ha], Bf[cb][nt][hb], c, 0, 0, 0);''',
    '''#if RVDD_XP & 1
                        acc[nt][mt] = c;
                        asm volatile("" : "+v"(acc[nt][mt]) : "v"(Af[cb][mt][ha]), "v"(Bf[cb][nt][hb]));
#else
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[cb][mt][ha], Bf[cb][nt][hb], c, 0, 0, 0);
#endif''')
rep('''                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);''',
    '''#if !(RVDD_XP & 2)
                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);
#endif''')
# 128 (round 5): the input map is stored SPLIT (per pixel six blocks of [hi of 8 channels, 16 B | lo of 8 channels, 16 B]; the harness
# converts its input): a loaded 16-B piece goes to its plane as it is, one ds_write_b128 -- the real consumer side of such a map
rep('''                        split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);
                        asm volatile("" : "+v"(shi[r0]), "+v"(slo[r0]));''',
    '''#if RVDD_XP & 128
                        asm volatile("" : "+v"(pre[r0]));
#elif !(RVDD_XP & 2)
                        split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);
                        asm volatile("" : "+v"(shi[r0]), "+v"(slo[r0]));
#endif''')
rep('''#pragma unroll
        for (int r0 = 0; r0 < G::NR; ++r0)
            if (ld_thread && G::RPR * r0 + rp < IH) {
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S) = shi[r0];
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S + G::PLANE) = slo[r0];
            }''', '''#pragma unroll
        for (int r0 = 0; r0 < G::NR; ++r0)
            if (ld_thread && G::RPR * r0 + rp < IH) {
#if RVDD_XP & 128
                *(lds_f4*)(L + plane0 + (unsigned)((rp * IW + hx) * G::S + (part >> 1) * 16 + ((part & 1) ? G::PLANE : 0)) + r0 * G::RPR * IW * G::S) = pre[r0];
#else
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S) = shi[r0];
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S + G::PLANE) = slo[r0];
#endif
            }''')
i = s.rindex('        write_tile();')
s = s[:i] + '''#if !(RVDD_XP & 2)
        write_tile();
#endif''' + s[i + len('        write_tile();'):]
rep('''    auto store_prev = [&](int i) {
        bstore(''', '''    auto store_prev = [&](int i) {
#if RVDD_XP & 4
        asm volatile("" : : "v"(outv[i]));
        return;
#endif
        bstore(''')
rep('''    auto read_frags = [&](int buf, int j) {''', '''    auto read_frags = [&](int buf, int j) {
#if RVDD_XP & 8
        if (j >= 2) return;
#endif''')
# 16: the epilogue's arithmetic (scale / bias / activation / maxima) replaced by a copy of the accumulators
rep('''        f32x4 v[2][MT];
        float m3 = 0.f;''', '''#if RVDD_XP & 16
        {
            constexpr int NO = (EPI == EPI_POOL ? 1 : 2) * MT;
            for (int i = 0; i < NO; ++i) outv[i] = acc[i / MT][i % MT];
            so_prev[0] = so[0];
            so_prev[1] = so[1];
            orr_prev = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)cur.b * a.Hout * a.Wout * kF), 0, out_bytes, 0x00020000);
            goto after_epilogue;
        }
#endif
        f32x4 v[2][MT];
        float m3 = 0.f;''')
rep('''        STAMP(4);
        gsync();               // every wave of the group has read its last fragment''', '''#if RVDD_XP & 16
    after_epilogue:
#endif
        STAMP(4);
        gsync();               // every wave of the group has read its last fragment''')
rep('''    auto gsync = [&]() {
        if constexpr (NGRP == 1) {''', '''    auto gsync = [&]() {
#if RVDD_XP & 32
        return;
#endif
        if constexpr (NGRP == 1) {''')
# 64 (round 5): the PRODUCER's share of a pre-split map -- every 16-B piece the epilogue hands to the stores is split into its f16
# halves (split4) and the halves of two neighbouring channel groups change lanes (two v_permlane16_swap), so that a lane again
# stores one 16-B piece ([hi of 8 channels] or [lo of 8 channels]); with bit 2 (no input side in the consumer) this is the UPPER
# bound of "maps stored split, halo by LDS-DMA" (VERDICT r4 item 3): the DMA itself and its addresses are still missing
rep('''                    outv[nt * MT + mt] = x;''', '''#if RVDD_XP & 64
                    {
                        u32x2 hh_, ll_;
                        split4<false>(x, 1.f, hh_, ll_);
                        auto s0_ = __builtin_amdgcn_permlane16_swap(ll_[0], hh_[0], false, false);
                        auto s1_ = __builtin_amdgcn_permlane16_swap(ll_[1], hh_[1], false, false);
                        x = __builtin_bit_cast(f32x4, u32x4{s0_[0], s1_[0], s0_[1], s1_[1]});
                    }
#endif
                    outv[nt * MT + mt] = x;''')
# 256 (round 5): no filter-bank copy at the top of the kernel (what a launch of one tile per workgroup pays for it)
rep('''        for (int i = wave; i < NP; i += NTHREADS / 64) {
            int k = i + rot;''', '''        for (int i = wave; i < NP && !(RVDD_XP & 256); i += NTHREADS / 64) {
            int k = i + rot;''')
# 512 / 1024 (round 6): a PART of the fused feature warp in the halo fetch -- every halo piece gathered from 4 (512) or 8
# (1024) sixteen-byte loads of a 4-wide footprint (rows y, y + 1) and combined with 4 / 8 packed-f32 FMAs one chunk later (the loads of one
# round in flight at a time: 16 / 32 registers, what the kernel has free).  The bicubic warp needs 16 loads and 16 FMAs per piece plus its
# taps' table; timing only (wrong results), no table: what a quarter / a half of the gather costs the L0 second pass.
rep('''    auto fetch_round = [&](const Src& q, int r0) { pre[r0] = bload(q.r, q.base + (unsigned)(r0 * G::RPR * a.W * (CIN * 4))); };''',
    '''#if RVDD_XP & (512 | 1024)
    constexpr int NGTH = (RVDD_XP & 1024) ? 8 : 4;
    f32x4 gth[NGTH];
    auto fetch_round = [&](const Src& q, int r0) {
#pragma unroll
        for (int k = 0; k < NGTH; ++k) gth[k] = bload(q.r, q.base + (unsigned)((r0 * G::RPR + (k >> 2)) * a.W * (CIN * 4) + (k & 3) * (CIN * 4)));
    };
    auto combine_round = [&](int r0) {
        f32x4 s_ = gth[0] * 0.53125f;
#pragma unroll
        for (int k = 1; k < NGTH; ++k) s_ = __builtin_elementwise_fma(gth[k], f32x4{0.0625f * k, 0.0625f * k, 0.0625f * k, 0.0625f * k}, s_);
        pre[r0] = s_;
    };
#else
    auto fetch_round = [&](const Src& q, int r0) { pre[r0] = bload(q.r, q.base + (unsigned)(r0 * G::RPR * a.W * (CIN * 4))); };
#endif''')
rep('''#if !(RVDD_XP & 2)
                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);
#endif''',
    '''#if RVDD_XP & (512 | 1024)
                if (j >= SH + 1 && j - SH - 1 < G::NR) combine_round(j - SH - 1);
                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);
#elif !(RVDD_XP & 2)
                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);
#endif''')
rep('''            for (int r0 = 0; r0 < G::NR; ++r0) fetch_round(q, r0);
#pragma unroll
            for (int r0 = 0; r0 < G::NR; ++r0) split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);''',
    '''            for (int r0 = 0; r0 < G::NR; ++r0) {
                fetch_round(q, r0);
#if RVDD_XP & (512 | 1024)
                combine_round(r0);
#endif
            }
#pragma unroll
            for (int r0 = 0; r0 < G::NR; ++r0) split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);''')
s = s.replace('#include "rvdd_internal.h"', '#ifndef RVDD_XP\n#define RVDD_XP 0\n#endif\n#include "rvdd_internal.h"', 1)
out = sys.argv[1]
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
open(out, "w").write(s)
print("wrote", out)
