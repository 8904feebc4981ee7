// conv_bf16_v2.hip — the trunk's 3x3 / stride 1 / pad 1 convolutions on bf16 activations, forward and dgrad, rebuilt as a
// PERSISTENT, LDS-DMA-fed implicit GEMM (round 3).  Replaces nn.Conv2d at normalization.py:41-42,73-74 (mlp_gamma_o |
// mlp_beta_o), sftmd_arch.py:811-820 (the DGB convolutions), :128-146, :860-866, :891-908 on the bf16 path.
//
// Why a second kernel.  k_conv3x3_bf16 (conv_bf16_mfma.hip) stages every 32-channel chunk through registers with two
// barriers per chunk and lives for ONE tile: 0.37 of the bf16 MFMA peak on 128 -> 128, the ceiling of that structure
// (cdna_hip_programming.md, "the step-3 structure's ~900 TF ceiling").  These convolutions sit 1.5x above the machine's
// FLOP/byte ridge at 128 channels and BELOW it at 64, so the kernel is built around keeping HBM streaming while the
// matrix pipe runs:
//   * a workgroup (4 waves, 8 x 32 output pixels x 32*NT output channels, two workgroups per CU) is persistent: it walks a
//     list of (pixel tile, channel slice) items and prefetches the NEXT item's first operands during the current item's
//     last K-steps, so that no item starts with an exposed HBM round trip;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The 10 x 34
//     halo tile of a 32-channel chunk (one DMA piece per thread per K-step, double-buffered) serves nine K-steps (taps);
//     the kernel slice of a K-step ([32*NT output channels][32 channels], 4-8 KB, L2-resident) runs two steps ahead in a
//     ring of three;
//   * ONE raw s_barrier per K-step (16 or 8 MFMAs per wave), preceded by a COUNTED s_waitcnt vmcnt(N) - never 0 in the
//     loop: the transfers of the next two steps stay in flight across it.  The DMA is issued from inline asm because
//     behind the builtin hipcc drains vmcnt(0) before every LDS read; the kernel therefore counts by hand, and every wave
//     issues exactly the same number of vector-memory operations per step (EXEC full; halo pixels outside the image and
//     the padding pieces of the 24 KB halo buffer read a zero page instead of being skipped).
//   * both LDS images are lane-linear, as LDS-DMA writes them, and XOR-swizzled through the SOURCE address: 16-byte piece
//     s of 64-byte row R holds channels 8 (s ^ ((R >> 2) & 3)), so that the 16 lanes of every ds_read_b128 group (rows
//     distinct mod 16) cover the 16 slots of a 256-byte bank row exactly once.
//   * the MFMA operands are swapped (D^T = W^T X^T): a lane ends up with 4 CONSECUTIVE output channels of one pixel per
//     register quad, the epilogue bounces them through LDS with 8-byte writes (bf16) and stores 16 bytes per lane, 8 full
//     128-byte lines per wave instruction.  Bias is preloaded into the accumulators from an LDS copy.
// K order: chunk -> tap -> 16 channels; fp32 accumulation in the matrix core (the same arithmetic as the first kernel in a
// different summation order).  Everything outside Cin % 32 == 0, Cout % 64 == 0, PixelShuffle in {1, 2} stays on the first
// kernel.
#include "bf16.h"
#include "conv_kernels.h"

#define V2_HW 34                 // halo tile width (32 + 2)
#define V2_EPITCH 144            // epilogue scratch: bytes per pixel row (128 + 16: 16-byte aligned, rows spread over banks)

// Two shapes of workgroup.  NWV = 4: 8 x 32 pixels, two workgroups per CU (waves of different workgroups share a SIMD and
// interleave freely), kernel slices two steps ahead in a ring of three.  NWV = 8: 16 x 32 pixels, ONE workgroup per CU: a
// kernel slice - the dominant L2 -> LDS traffic, 295 KB per item at 128 -> 128 - serves twice the pixels, and the LDS of the
// whole CU holds a ring of NINE slices (slot = tap) running FIVE steps ahead: per CU ~48 KB of DMA in flight instead of ~40
// for half the bytes per FLOP (measured with everything but the DMA and the stores compiled out, the 4-wave form streams at
// 3 TB/s: latency x bytes in flight, not the matrix pipe, bounded it).
template <int NT, int NWV, int DBG>
struct V2Geom {
    static constexpr int NTHR = 64 * NWV, TH = 2 * NWV, HPIX = (TH + 2) * V2_HW;
    static constexpr int NHP = (HPIX * 4 + NTHR - 1) / NTHR;      // halo pieces per thread per chunk: 6 (340 px) / 5 (612 px)
    static constexpr int HBYTES = NHP * NTHR * 16;                  // 24 576 / 40 960: whole wave-instructions, EXEC full
    static constexpr int NTILE = 32 * NT, SLAB = NTILE * 64, WPIECES = SLAB / 16;
    static constexpr int R = NWV == 8 ? 9 : 3, D = NWV == 8 ? 5 : 2;
    // PIPE (the 8-wave form): the operand reads run two 2-MFMA units ahead of the matrix instructions, ACROSS the step
    // barrier - the first fragments of step t + 1 are read at the end of step t, so the barrier of step t must already
    // cover the slice of step t + 1: one step less of DMA run-ahead (waitn sums D - 2 steps instead of D - 1)
    static constexpr bool PIPE = NWV == 8;
    static constexpr int DW = PIPE ? D - 1 : D;
    static constexpr int NWP = WPIECES >= NTHR ? WPIECES / NTHR : 1;   // kernel-slice pieces a thread issues when it is its turn
    static constexpr bool SPLIT = WPIECES < NTHR;                   // 8 waves, 4 KB slice: wave group (tap & 1) fetches it
    // vector-memory operations a thread of wave group g issues in the step of tap t: halo piece t, the slice of tap t + D
    static constexpr int own(int g, int t) {
        return ((t < NHP && !(DBG & 2)) ? 1 : 0) + ((DBG & 1) ? 0 : (SPLIT ? (((((t + D) % 9) & 1) == g) ? 1 : 0) : NWP));
    }
    // ... and in the DW - 1 steps before the step of tap t: everything older (the slice of this step - PIPE: and of the next -
    // issued D steps ago, and this chunk's halo pieces) has landed once all but that many of the wave's operations are done
    static constexpr int waitn(int g, int t) {
        int n = 0;
        for (int k = 1; k < DW; ++k) n += own(g, (t - k + 18) % 9);
        return n;
    }
};

DASR_DEVICE_CONST __attribute__((aligned(64))) unsigned v2_zero_page[64] = {0};

struct ConvV2Args {
    const bf16_t* x;         // [B,H,W,Cin]
    const bf16_t* w;         // packed bf16 kernel [2][9][.][.] (HWIO, then [tap][co][ci]); for dgrad Cin / Cout are swapped
    const float* bias;       // [Cout] or null
    const bf16_t* residual;  // [B,H,W,Cout] or null
    bf16_t* y;
    int B, H, W, Cin, Cout;
    int act, ps_r, accumulate;
    int tiles_x, tiles_y, nsl, nitems, Q, G8;    // item = (pixel tile, N slice); Q items per XCD, G8 workgroups per XCD
};

template <int N>
__device__ __forceinline__ void v2_wait_vm() {
#ifndef DASR_HIPEMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// DBG: timing experiments only (wrong results), reachable through dasr_set_conv_bf16_impl(2 | flags << 13) when the library is
// built with -DDASR_V2_DEBUG: 1 = no kernel-slice DMA in the loop, 2 = no halo DMA in the loop, 4 = no MFMA, 8 = no output
// stores, 16 = no per-step wait / barrier, 32 = no operand LDS reads.
// the same with the count as a value that constant-folds after unrolling (the tap loop's index is not a constant expression)
__device__ __forceinline__ void v2_wait_vm_n(int n) {
    switch (n) {
#define V2_W(N) case N: v2_wait_vm<N>(); break;
        V2_W(0) V2_W(1) V2_W(2) V2_W(3) V2_W(4) V2_W(5) V2_W(6) V2_W(7) V2_W(8) V2_W(9) V2_W(10) V2_W(11) V2_W(12)
        V2_W(13) V2_W(14) V2_W(15) V2_W(16)
#undef V2_W
        default: v2_wait_vm<0>(); break;
    }
}

template <int NT, int WMODE, int NWV, int DBG = 0>
__global__ void __launch_bounds__(64 * NWV, 2) k_conv3x3_bf16_v2(ConvV2Args a) {
    DASR_DYN_SMEM(smem);
    typedef V2Geom<NT, NWV, DBG> G;
    constexpr int NTILE = G::NTILE, SLAB = G::SLAB, NTHR = G::NTHR, NHP = G::NHP, HBYTES = G::HBYTES, R = G::R, D = G::D;
    constexpr int NW = G::NWP;
    char* const sH = smem;                         // [2][HBYTES]
    char* const sW = smem + 2 * HBYTES;            // [R][SLAB]
    float* const sBias = (float*)(sW + R * SLAB);  // [Cout]
    const int tid = threadIdx.x, lane = tid & 63, wv = DASR_UNIFORM((int)(tid >> 6));
    const int li = lane & 31, lh = lane >> 5;
    const int grp = wv >> 2;                       // (8 waves) the half of the workgroup this wave belongs to
    // LDS byte addresses of the DMA destinations, as integers from ONE pointer conversion (each generic -> LDS pointer cast
    // costs a null check in the scalar unit)
    const dasr_lds_addr_t ldsH = DASR_LDS_ADDR(sH) + 1024 * wv, ldsW = DASR_LDS_ADDR(sW) + 1024 * (G::SPLIT ? (wv & 3) : wv);

    // ---- work list of this workgroup: items [ibeg, iend) belong to its XCD (neighbouring tiles share halo rows in that
    // XCD's L2), dealt round-robin to the XCD's G8 workgroups
    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int ibeg = xcd * a.Q;
    const int iend = ibeg + a.Q < a.nitems ? ibeg + a.Q : a.nitems;
    int item = ibeg + jwg;
    if (item >= iend) return;                      // (whole workgroup; before any barrier or DMA)
    const int NC = a.Cin >> 5;
    const int pixb = a.Cin * 2, rowb = a.W * pixb;
    const size_t sampb = (size_t)a.H * rowb;
    const char* const zp = (const char*)v2_zero_page;

    for (int i = tid; i < a.Cout; i += NTHR) sBias[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();                               // (no DMA in flight yet: a plain barrier with its LDS fence)

    // kernel-slice pieces: piece v of a thread = slot (tid & 3) of slice row nl = (tid >> 2) + NTHR/4 v (split: the issuing
    // half's 256 threads cover the 64 rows)
    int wrel[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) {
        const int nl = G::SPLIT ? ((tid & 255) >> 2) : (tid >> 2) + (NTHR / 4) * v;
        wrel[v] = (nl * a.Cin + 8 * ((tid & 3) ^ ((nl >> 2) & 3))) * 2;
    }
    const char* const wbase = (const char*)(WMODE == 0 ? a.w + (size_t)9 * a.Cin * a.Cout : a.w);
    const int tapb = a.Cout * a.Cin * 2;

    // item state (wave-uniform): tile origin, sample base, slice
    int x0, y0, n0, bb;
    auto decode = [&](int it, int& ox0, int& oy0, int& on0, int& ob) {
        const int ns = it % a.nsl, pt = it / a.nsl;
        const int tile = pt % (a.tiles_x * a.tiles_y);
        ob = pt / (a.tiles_x * a.tiles_y);
        ox0 = (tile % a.tiles_x) * 32;
        oy0 = (tile / a.tiles_x) * G::TH;
        on0 = ns * NTILE;
    };
    decode(item, x0, y0, n0, bb);
    // halo source state of the item whose chunks are being fetched (the current item, or the next one during the last chunk)
    int hoff[NHP];
    unsigned hok = 0;
    const char* hxb;
    // piece u of a thread = 16-byte slot (tid & 3) of halo pixel P = (tid >> 2) + NTHR/4 u.  Its (row, column) decode is redone
    // per item from an opaque copy of tid (a dozen integer operations per piece, once per ~10 us item): hoisted out of the
    // item loop it would hold 18 registers for the whole kernel, which the 128 accumulators do not leave room for.
    auto halo_setup = [&](int fx0, int fy0, int fb, bool real) {
        hxb = (const char*)a.x + (size_t)fb * sampb;
        const int org = (fy0 - 1) * rowb + (fx0 - 1) * pixb;
        int t = tid;
#ifndef DASR_HIPEMU
        asm volatile("" : "+v"(t));
#endif
        hok = 0;
#pragma unroll
        for (int u = 0; u < NHP; ++u) {
            const int P = (t >> 2) + (NTHR / 4) * u;
            const int pr = P / V2_HW, pc = P - pr * V2_HW;
            hoff[u] = org + pr * rowb + pc * pixb + 16 * ((t & 3) ^ ((P >> 2) & 3));
            const bool ok = real && P < G::HPIX && (unsigned)(fy0 - 1 + pr) < (unsigned)a.H && (unsigned)(fx0 - 1 + pc) < (unsigned)a.W;
            hok |= ok ? (1u << u) : 0u;
        }
    };
    auto halo_issue = [&](int u, int cc, int buf) {
        const char* src = ((hok >> u) & 1u) ? hxb + hoff[u] + 64 * cc : zp;
        if (DBG & 128)      // timing experiment: the same bytes from a contiguous (chunk-planar) image
            src = hxb + (size_t)(((y0 * a.W + x0) * 64 + cc * a.H * a.W * 64 + (tid + NTHR * u) * 16) & 0x3ffffff);
        DASR_GLDS16(src, ldsH + buf * HBYTES + 1024 * NWV * u);
    };
    // kernel slice of K-step (chunk cc, tap) of channel slice fn0 -> ring slot
    auto w_issue = [&](int cc, int tap, int fn0) {
        if (G::SPLIT && (tap & 1) != grp) return;             // (wave-uniform) the other half of the workgroup fetches this one
        const int tsrc = WMODE == 0 ? tap : 8 - tap;
        const char* src = wbase + (size_t)tsrc * tapb + (size_t)(fn0 * a.Cin + 32 * cc) * 2;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const char* s2 = src + wrel[v];
            if (DBG & 64)       // timing experiment: the slice as one contiguous block (a slab-major packed kernel)
                s2 = wbase + (size_t)((tsrc * (a.Cin >> 5) + cc) * SLAB) + (tid + NTHR * v) * 16;
            DASR_GLDS16(s2, ldsW + (tap % R) * SLAB + 1024 * NWV * v);
        }
    };

    // operand read offsets: A row P = (2 wv + m + dy) * 34 + li + dx, B row = 32 n + li; piece (2q + lh) ^ ((row >> 2) & 3)
    const int Pl = 2 * wv * V2_HW + li;
    const int boff = li * 64 + ((lh ^ ((li >> 2) & 3)) << 4);

    // ---- prologue: the first item's first chunk and first D kernel slices
    int par = 0;
    halo_setup(x0, y0, bb, true);
#pragma unroll
    for (int u = 0; u < NHP; ++u) halo_issue(u, 0, 0);
#pragma unroll
    for (int t = 0; t < D; ++t) w_issue(0, t, n0);

    for (;;) {
        const int nitem = item + a.G8;
        const bool has_next = nitem < iend;
        int nx0 = x0, ny0 = y0, nn0 = n0, nb = bb;
        if (has_next) decode(nitem, nx0, ny0, nn0, nb);

        bf16x8 pA[2][2], pB[3];         // (PIPE) operand fragments in flight: A of both 16-channel halves, a ring of three B
        // accumulators start at the bias: a lane's register quad g of acc[.][n] = channels n0 + 32 n + 8 g + 4 lh .. + 3
        f32x16 acc[2][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(sBias + n0 + 32 * n + 8 * g + 4 * lh);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][n][4 * g] = bv.x; acc[m][n][4 * g + 1] = bv.y;
                    acc[m][n][4 * g + 2] = bv.z; acc[m][n][4 * g + 3] = bv.w;
                }
            }

        for (int cc = 0; cc < NC; ++cc) {
            const bool last = cc == NC - 1;
            if (last) halo_setup(nx0, ny0, nb, has_next);       // from here on the halo prefetch belongs to the next item
            const int fcc = last ? 0 : cc + 1;
            char* const hb = sH + par * HBYTES;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // the slice of this step (issued D steps ago) and this chunk's halo pieces have landed once all but the
                // operations this wave issued in the last D - 1 steps are done; then every wave's have (barrier)
                if (!(DBG & 16)) {
                    if (G::waitn(0, tap) == G::waitn(1, tap) || grp == 0) v2_wait_vm_n(G::waitn(0, tap));
                    else                                                  v2_wait_vm_n(G::waitn(1, tap));
                    DASR_RAW_BARRIER();
                }
                // prefetch: one halo piece of the next chunk, the kernel slice D steps ahead (its ring slot was last read
                // R - D >= 1 steps ago: every wave is past those reads, it has arrived at this step's barrier)
                if (tap < NHP && !(DBG & 2)) halo_issue(tap, fcc, par ^ 1);
                if (!(DBG & 1)) {
                    if (tap + D < 9) w_issue(cc, tap + D, n0);
                    else             w_issue(fcc, tap + D - 9, last ? nn0 : n0);
                }
                // compute.  The A read addresses (row base + swizzled piece: not affine in the tap, a handful of integer
                // operations each) are derived from an OPAQUE per-step copy of the lane's pixel index: left visible, the
                // 36 (tap, row, half) offsets are loop-invariant and hipcc keeps them all in registers for the whole kernel,
                // which with 128 accumulators means spills.
                const int dy = tap / 3, dx = tap - 3 * dy;
                const char* const wb = sW + (tap % R) * SLAB;
                int Pq = Pl;
#ifndef DASR_HIPEMU
                asm volatile("" : "+v"(Pq));
#endif
                if (G::PIPE) {
                    // Units u = 0 .. 2 NT - 1 of the step: unit (q = u / NT, n = u % NT) = one B fragment and its two MFMAs
                    // (tile rows m = 0, 1).  B fragments live in a ring of three, read two units ahead; the A fragments
                    // of a 16-channel half are read two units before its first use; the last two units of a step read the
                    // first fragments of the NEXT step (its slice and halo chunk are covered by this step's barrier).  The
                    // first step of an item reads its own (nothing is carried across the epilogue).
                    constexpr int U = 2 * NT;
                    const int ntap = (tap + 1) % 9;
                    const int ndy = ntap / 3, ndx = ntap - 3 * ndy;
                    const char* const nhb = tap == 8 ? sH + (par ^ 1) * HBYTES : hb;
                    const char* const nwb = sW + (ntap % R) * SLAB;
                    const int ph = (U * tap) % 3;                       // ring phase of this step's unit 0
                    const bool carry = !(last && tap == 8);             // (wave-uniform) a next step exists in this item
                    auto rdA = [&](const char* hbuf, int ddy, int ddx, int q, bf16x8 (&A)[2]) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int P = Pq + (m + ddy) * V2_HW + ddx;
                            A[m] = *(const bf16x8*)(hbuf + P * 64 + ((((2 * q + lh) ^ (P >> 2)) & 3) << 4));
                        }
                    };
                    auto rdB = [&](const char* wbuf, int u, bf16x8& B) {
                        B = *(const bf16x8*)(wbuf + (u % NT) * 2048 + (boff ^ (32 * (u / NT))));
                    };
                    if (cc == 0 && tap == 0) {
                        rdA(hb, dy, dx, 0, pA[0]);
                        rdB(wb, 0, pB[ph % 3]);
                        rdB(wb, 1, pB[(ph + 1) % 3]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (u + 2 < U) rdB(wb, u + 2, pB[(ph + u + 2) % 3]);
                        else if (carry) rdB(nwb, u + 2 - U, pB[(ph + u + 2) % 3]);
                        if (u == (NT >= 2 ? NT - 2 : 0)) rdA(hb, dy, dx, 1, pA[1]);
                        if (u == U - 2 && carry) rdA(nhb, ndy, ndx, 0, pA[0]);
                        // pin the interleave: this unit's reads (1 or 3), then its two MFMAs - left alone, the scheduler
                        // pulls the reads of several units up front and the fragments in flight overflow the register file
                        if (u == (NT >= 2 ? NT - 2 : 0) || u == U - 2) DASR_SCHED_GROUP(0x100, 3);
                        else                                           DASR_SCHED_GROUP(0x100, 1);
                        DASR_SCHED_GROUP(0x008, 2);
                        const int q = u / NT, n = u % NT;
                        if (!(DBG & 4)) {
                            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pB[(ph + u) % 3], pA[q][0], acc[0][n], 0, 0, 0);
                            acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pB[(ph + u) % 3], pA[q][1], acc[1][n], 0, 0, 0);
                        } else {
                            acc[0][n][0] += dasr_bf2f(pB[(ph + u) % 3][0]) + dasr_bf2f(pA[q][0][0]);
                            acc[1][n][0] += dasr_bf2f(pB[(ph + u) % 3][7]) + dasr_bf2f(pA[q][1][7]);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bf16x8 Af[2], Bf[NT];
                    if (DBG & 32) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            Af[0][e] = Af[1][e] = dasr_f2bf((float)(tap + q));
#pragma unroll
                            for (int n = 0; n < NT; ++n) Bf[n][e] = dasr_f2bf((float)(cc + n));
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int P = Pq + (m + dy) * V2_HW + dx;
                            Af[m] = *(const bf16x8*)(hb + P * 64 + ((((2 * q + lh) ^ (P >> 2)) & 3) << 4));
                        }
#pragma unroll
                        for (int n = 0; n < NT; ++n) Bf[n] = *(const bf16x8*)(wb + n * 2048 + (boff ^ (32 * q)));
                    }
                    if (DBG & 4) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {       // keep the operands live without the matrix work
                            acc[0][n][0] += dasr_bf2f(Bf[n][0]) + dasr_bf2f(Af[0][0]);
                            acc[1][n][0] += dasr_bf2f(Bf[n][7]) + dasr_bf2f(Af[1][7]);
                        }
                        continue;
                    }
                    DASR_SETPRIO(1);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[n], Af[m], acc[m][n], 0, 0, 0);
                    DASR_SETPRIO(0);
                }
            }
            par ^= 1;
        }

        // ---- epilogue.  Scratch = the halo buffer of the chunk just finished (the next item's first chunk is landing in
        // the other one); every wave must be done reading it.
        DASR_RAW_BARRIER();
        char* const scr = sH + (par ^ 1) * HBYTES + wv * (32 * V2_EPITCH);
        const bool is_relu = a.act == DASR_ACT_RELU;
        const float slope = a.act == DASR_ACT_LRELU02 ? 0.2f : 1.f;
        const int wvalid = a.W - x0;
        const bool plain = a.ps_r == 1 && a.residual == nullptr && !a.accumulate;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gy = y0 + 2 * wv + m;
            if (plain) {
                // bias is in the accumulator: activation, round to bf16, 64 channels per pass through [pixel][64 + 8] bf16
#pragma unroll
                for (int h = 0; h < NT / 2; ++h) {
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const float o = acc[m][2 * h + nn][4 * g + t];
                                v[t] = is_relu ? fmaxf(o, 0.f) : fmaxf(o, o * slope);
                            }
                            const bf16x2_t p0 = dasr_f2bf2(v[0], v[1]), p1 = dasr_f2bf2(v[2], v[3]);
                            bf16x4 pk;
                            pk[0] = p0[0]; pk[1] = p0[1]; pk[2] = p1[0]; pk[3] = p1[1];
                            *(bf16x4*)(scr + li * V2_EPITCH + (32 * nn + 8 * g + 4 * lh) * 2) = pk;
                        }
                    DASR_WAVE_SYNC();
                    if (gy < a.H) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int v = lane + 64 * u, pix = v >> 3, j8 = v & 7;
                            const bf16x8 ov = *(const bf16x8*)(scr + pix * V2_EPITCH + 16 * j8);
                            if (pix < wvalid && (!(DBG & 8) || ov[0] == (bf16_t)12345.f))
                                *(bf16x8*)(a.y + (((size_t)bb * a.H + gy) * a.W + x0 + pix) * a.Cout + n0 + 64 * h + 8 * j8) = ov;
                        }
                    }
                    DASR_WAVE_SYNC();
                }
            } else {
                // residual / accumulate / PixelShuffle(2): 32 channels per pass through [pixel][32 + 4] fp32
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 pk = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                        *(f32x4*)(scr + li * V2_EPITCH + (8 * g + 4 * lh) * 4) = pk;
                    }
                    DASR_WAVE_SYNC();
                    if (gy < a.H) {
                        if (a.ps_r == 1) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int v = lane + 64 * u, pix = v >> 2, cg = v & 3;
                                if (pix >= wvalid) continue;
                                const float4 lo = *(const float4*)(scr + pix * V2_EPITCH + 32 * cg);
                                const float4 hi = *(const float4*)(scr + pix * V2_EPITCH + 32 * cg + 16);
                                float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                                const size_t idx = (((size_t)bb * a.H + gy) * a.W + x0 + pix) * a.Cout + n0 + 32 * n + 8 * cg;
                                if (a.residual) {
                                    const bf16x8 rv = *(const bf16x8*)(a.residual + idx);
#pragma unroll
                                    for (int t = 0; t < 8; ++t) o[t] += dasr_bf2f(rv[t]);
                                }
#pragma unroll
                                for (int t = 0; t < 8; ++t) o[t] = is_relu ? fmaxf(o[t], 0.f) : fmaxf(o[t], o[t] * slope);
                                if (a.accumulate) {
                                    const bf16x8 av = *(const bf16x8*)(a.y + idx);
#pragma unroll
                                    for (int t = 0; t < 8; ++t) o[t] += dasr_bf2f(av[t]);
                                }
                                bf16x8 ov;
#pragma unroll
                                for (int t = 0; t < 8; ++t) ov[t] = dasr_f2bf(o[t]);
                                *(bf16x8*)(a.y + idx) = ov;
                            }
                        } else {
                            // PixelShuffle(2): out[b, 2gy+i, 2gx+j, c] = conv[b, gy, gx, 4c + 2i + j]; this pass holds 8 values
                            // of c for each of the 4 sub-pixels of 32 pixels: one 16-byte store per (pixel, sub-pixel)
                            const int Cq = a.Cout / 4;
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int v = lane + 64 * u, j = v & 1, pix = (v >> 1) & 31, i = v >> 6;
                                if (pix >= wvalid) continue;
                                const float* sp = (const float*)(scr + pix * V2_EPITCH) + 2 * i + j;
                                bf16x8 ov;
#pragma unroll
                                for (int t = 0; t < 8; ++t) {
                                    const float o = sp[4 * t];
                                    ov[t] = dasr_f2bf(is_relu ? fmaxf(o, 0.f) : fmaxf(o, o * slope));
                                }
                                const size_t idx = (((size_t)bb * a.H * 2 + 2 * gy + i) * ((size_t)a.W * 2) + 2 * (x0 + pix) + j) * Cq +
                                                   (n0 + 32 * n) / 4;
                                *(bf16x8*)(a.y + idx) = ov;
                            }
                        }
                    }
                    DASR_WAVE_SYNC();
                }
            }
        }
        if (!has_next) break;
        item = nitem; x0 = nx0; y0 = ny0; n0 = nn0; bb = nb;
    }
    // the (unused) prefetches of the last item must have landed before this workgroup's LDS can be handed to another one
    v2_wait_vm<0>();
}

// ------------------------------------------------------------------------------------------ host side
// bits 0-1: 0 = v2 where it applies; 1 = first kernel everywhere (A/B measurements, tests); 2 = v2 with ONE workgroup per XCD
// (tests: every workgroup then walks a long item list, which small test shapes would not do with 32-64 workgroups per XCD).
// bits 4-5: 0 = tile height by problem size, 1 = 8-row tiles / 4 waves, 2 = 16-row tiles / 8 waves.
static int g_conv_bf16_impl = 0;
static int g_conv_bf16_dbg = 0;
static int g_conv_bf16_v2_launches = 0;
extern "C" int dasr_set_conv_bf16_impl(int impl) {
    const int dbg = impl >> 13;
    impl &= 8191;
#ifndef DASR_V2_DEBUG
    if (dbg != 0) return DASR_E_UNSUPPORTED;
#endif
    if (impl < 0 || (impl & 3) > 2 || ((impl >> 4) & 3) > 2 || (impl & 12) || (impl >> 13)) return DASR_E_UNSUPPORTED;   // (+ 64 / + 128: see sw_launch, conv_split_bf16.hip)
    g_conv_bf16_impl = impl;
    g_conv_bf16_dbg = dbg;
    return DASR_OK;
}
extern "C" int dasr_get_conv_bf16_impl(void) { return g_conv_bf16_impl; }
extern "C" int dasr_conv_bf16_v2_launches(void) { return g_conv_bf16_v2_launches; }

// geometry in the kernel's terms: Cin = reduction channels, Cout = produced channels (dgrad swaps the layer's two)
bool conv_bf16_v2_supported(int H, int W, int Cin, int Cout, int ps_r, bool has_residual, bool accumulate) {
    if ((g_conv_bf16_impl & 3) == 1) return false;
    if ((Cin % 32) != 0 || (Cout % 64) != 0 || Cout > 1024) return false;
    if (ps_r != 1 && !(ps_r == 2 && !has_residual && !accumulate)) return false;
    const size_t cmax = Cin > Cout ? Cin : Cout;
    return (size_t)H * W * cmax * sizeof(bf16_t) < ((size_t)1 << 31) && (size_t)9 * Cin * Cout * sizeof(bf16_t) < ((size_t)1 << 31);
}

template <int NT, int WMODE, int NWV>
static int launch_v2(ConvV2Args& a, void* stream) {
    typedef V2Geom<NT, NWV, 0> G;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    a.nsl = a.Cout / G::NTILE;
    a.nitems = a.tiles_x * a.tiles_y * a.B * a.nsl;
    a.Q = (a.nitems + 7) / 8;
    const int per_xcd = NWV == 8 ? 32 : 64;          // one 512-thread or two 256-thread workgroups per CU
    a.G8 = a.Q < per_xcd ? a.Q : per_xcd;
    if ((g_conv_bf16_impl & 3) == 2) a.G8 = 1;
    ++g_conv_bf16_v2_launches;
    const size_t lds = 2 * (size_t)G::HBYTES + (size_t)G::R * G::SLAB + sizeof(float) * (size_t)a.Cout;
    const dim3 grid(8 * a.G8);
#ifdef DASR_V2_DEBUG
#define V2_DBG_CASE(DB)                                                                                        \
    if (g_conv_bf16_dbg == DB) {                                                                              \
        DASR_LAUNCH((k_conv3x3_bf16_v2<NT, 0, NWV, DB>), grid, dim3(G::NTHR), lds, stream, a);                \
        DASR_RETURN_LAUNCH_STATUS();                                                                          \
    }
    V2_DBG_CASE(1) V2_DBG_CASE(2) V2_DBG_CASE(3) V2_DBG_CASE(4) V2_DBG_CASE(8) V2_DBG_CASE(16) V2_DBG_CASE(32) V2_DBG_CASE(36)
    V2_DBG_CASE(19) V2_DBG_CASE(12) V2_DBG_CASE(64) V2_DBG_CASE(128) V2_DBG_CASE(192) V2_DBG_CASE(100) V2_DBG_CASE(228)
#endif
    DASR_LAUNCH((k_conv3x3_bf16_v2<NT, WMODE, NWV>), grid, dim3(G::NTHR), lds, stream, a);
    DASR_RETURN_LAUNCH_STATUS();
}

template <int WMODE>
static int launch_conv_bf16_v2(ConvV2Args& a, void* stream) {
    const int NT = (a.Cout % 128) == 0 ? 4 : 2;
    // 16-row tiles (8 waves, one workgroup per CU) once there are at least two of them per CU; forced either way by the
    // test / measurement switch
    const int force = (g_conv_bf16_impl >> 4) & 3;
    const long items16 = (long)((a.W + 31) / 32) * ((a.H + 15) / 16) * a.B * (a.Cout / (32 * NT));
    const bool wide = force == 2 || (force == 0 && items16 >= 512);
    if (wide) return NT == 4 ? launch_v2<4, WMODE, 8>(a, stream) : launch_v2<2, WMODE, 8>(a, stream);
    return NT == 4 ? launch_v2<4, WMODE, 4>(a, stream) : launch_v2<2, WMODE, 4>(a, stream);
}

int conv_bf16_v2_fwd(const ConvGeom& g, const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* residual,
                     bf16_t* y, int act, int ps_r, void* stream) {
    ConvV2Args a{x, w, bias, residual, y, g.B, g.H, g.W, g.Cin, g.Cout, act, ps_r, 0, 0, 0, 0, 0, 0, 0};
    return launch_conv_bf16_v2<0>(a, stream);
}
int conv_bf16_v2_dgrad(const ConvGeom& g, const bf16_t* dconv, const bf16_t* w, bf16_t* dx, int accumulate, void* stream) {
    ConvV2Args a{dconv, w, nullptr, nullptr, dx, g.B, g.H, g.W, g.Cout, g.Cin, DASR_ACT_NONE, 1, accumulate, 0, 0, 0, 0, 0, 0};
    return launch_conv_bf16_v2<1>(a, stream);
}
