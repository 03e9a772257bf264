// qr_gemm_nt.hip -- the wide trailing update of the blocked Householder QR on gfx950 (MI355X), second generation.
//
// Replaces, for the tile-aligned interior of the wide update, the pair gemm_tn_kernel<4,4> / gemm_nn_w8_kernel of
// qr_kernels.hip (reference: trailingUpdateKernel qr.cu:335-465, host loop qr.c:255-293):
//
//     Wt = A2^T (V T)          (nt x nbp, "W transposed": the long dimension nt is the contiguous one)    gemm_tn_kernel (qr_kernels.hip), operands swapped
//     A2 -= V Wt^T             (mk x nt)                                                                  gemm_nt_kernel
//
// Storing W transposed makes BOTH operands of the update "row-fast": for a fixed k the 128 rows of a V tile and
// the 128 columns of a Wt tile are each 1 KiB of contiguous memory, i.e. exactly one global_load_lds_dwordx4 wave
// instruction (64 lanes x 16 B, LDS destination = wave-uniform base + lane*16).  So the tiles go HBM/L2 -> LDS
// directly: no staging VGPRs, no ds_write pass, no wait between a load and its LDS store.
//
// Fragment maps (v_mfma_f64_16x16x4_f64: A-operand lane l = Aop[p = l&15][k = l>>4], B-operand lane l =
// Bop[k = l>>4][q = l&15], D lane l reg r = D[p = (l>>4) + 4r][q = l&15]):
//   * rows go on p, columns on q, and the 16 rows of MFMA tile b of a wave are NOT consecutive: tile pair (2c, 2c+1)
//     covers wave rows 32c .. 32c+31 with   row(b, p) = 32c + 2p + (b & 1).
//     - a lane's A fragments for tiles 2c and 2c+1 are two adjacent doubles of the LDS row -> ONE ds_read_b128, and
//       the 16 lanes of a bank group read 256 contiguous bytes: conflict-free on the unpadded [k][128] image
//       (row stride 1024 B = 0 mod 256 B, which is what the b128 lane groups need);
//     - the D registers r of tiles 2c, 2c+1 in one lane are rows 32c + 2(l4 + 4r) + {0, 1} of one column: 16
//       contiguous bytes, so C moves with global_load/store_dwordx4 and the 4 lanes l4 of a column cover 64
//       contiguous bytes per instruction.
//   * columns: tile a of a wave holds columns 2q + a, so both B fragments of a lane are one ds_read_b128.
// Per 4-deep MFMA step a wave issues 3 ds_read_b128 for 8 MFMAs (the first generation: 6 ds_read_b64, plus 4
// ds_write_b128 and 4 global_load_dwordx4 per 16-deep tile).
// C enters through the accumulators and the product is subtracted by the MFMA's own negate-A modifier
// (blgp = 1 -> "neg:[1,0,0]"), so prologue and epilogue are 16 loads and 16 stores per lane, nothing else.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qr_device.h"
#include "qr_common.h"

#define NT_BM 128
#define NT_BN 128
#define NT_BK 16
#define NT_STAGE (2 * NT_BK * 128)          /* doubles per pipeline stage: A image [16][128] + B image [16][128] */

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*) (p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*) (p))

// (tile_of_block: workgroup -> tile map, XCD-aware; qr_common.h)

// ------------------------------------------------------------------------------------------------
// C (M x N, ldc) -= A (M x K, lda) * Bt (N x K, ldbt)^T      (NEG; else +=)
// M % 128 == N % 128 == K % 16 == 0, K >= 16, operands 16-byte aligned with even leading dimensions (host-checked).
// 512 threads = 8 waves as 2 (rows) x 4 (cols), wave tile 64 x 32 = 4 x 2 MFMA tiles, two LDS stages of 32 KiB.
// ------------------------------------------------------------------------------------------------
// (Round 3, measured and removed: a PERSISTENT form -- two workgroups per compute unit walking the tile list, the next tile's loads right
// behind the current tile's stores.  In-kernel stamps at 16384 x 16128 x 256: prologue 16.9 k, K loop 129.6 k, epilogue 5.6 k ticks = 64.7 us
// per workgroup while a slot turns over every 72.7 us; the persistent kernel nevertheless ran the update at 47.0 against 49.6 TFLOP/s in
// situ, and starting half of its workgroups half a tile late made it worse: profiles/r03_nt_variants.txt.)
// (Also measured and removed: k-tiles of 8 in FOUR LDS stages, tiles requested three ahead and the barrier no longer draining the load
// queue (s_waitcnt vmcnt(4)): 58.5 against 58.9 TFLOP/s isolated.  Stamps + placement (devtools/gemm_lab.cpp): the two workgroups of a
// compute unit get 0.86 of the matrix pipe while both are in their K loop and a lone one 0.68 -- 42 % of the time, while the other loads
// or stores its C tile or is being replaced; that per-tile cost (14.6 us of 72.7 at K = 256), not the K loop, is what K = 512 halves.)
// (And the last attempt at the per-tile cost: PERSISTENT workgroups that prefetch their next C tile into registers under the current
// tile's K loop (256 VGPRs at two waves per SIMD; one C load per k-tile behind an MFMA, the barrier waiting with s_waitcnt vmcnt(1) so
// that the newest C load stays in flight).  One 8-wave workgroup per CU: 57.9 TFLOP/s isolated (the non-persistent single workgroup:
// 51.7) -- all its waves meet every tile boundary at once.  Two independent 4-wave workgroups per CU on 128 x 64 tiles: 60.4 isolated,
// 50.6 against 49.6 in situ, the factorisation unchanged within noise (121.5 against 121.7 ms); 256 VGPRs + 64 B of scratch.  Removed:
// +2 % on this kernel did not pay for a second 170-line kernel.  profiles/r03_nt_variants.txt.)
// IL = 1: the K loop with its issue order spelled out -- one LDS read or tile load behind every other MFMA, the barrier in the middle of
// the last step's MFMAs with the next tile's first fragment reads behind it (see gemm_kloop_il in qr_gemm_tile.h for the measurement that
// led there); IL = 0: fragment reads one step ahead in groups of three, the four tile loads at the top of the tile
// CEIL (measurement only, results wrong): 1 = the C tile is neither loaded nor stored, 2 = nor are the operand tiles loaded (K loop on
// whatever the LDS holds): what the K loop alone / the matrix pipe alone deliver in the same launch mix (profiles/r05_nt_ceiling.txt)
template <bool NEG, int STAMP, int IL = 0, int CEIL = 0>
__global__ __launch_bounds__(512, 4) void gemm_nt_kernel(int M, int N, int K, const double* __restrict__ A, int lda,
                                                         const double* __restrict__ Bt, int ldbt,
                                                         double* __restrict__ C, int ldc, int gx, int gy, int gm,
                                                         unsigned long long* __restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    int tx, ty;
    tile_of_block(blockIdx.x, gx, gy, gm, tx, ty);
    const int i0 = tx * NT_BM, j0 = ty * NT_BN;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memtime();

    // --- tile loader: wave w brings k-rows w and w + 8 of both images; every instruction is 1 KiB contiguous
    const double* ga = A + (size_t) wave * lda + i0 + 2 * lane;
    const double* gb = Bt + (size_t) wave * ldbt + j0 + 2 * lane;
    const size_t a8 = (size_t) 8 * lda, b8 = (size_t) 8 * ldbt;
    auto issue = [&](int kt, int stage) {
        if (CEIL == 2) return;
        const double* pa = ga + (size_t) kt * NT_BK * lda;
        const double* pb = gb + (size_t) kt * NT_BK * ldbt;
        double* sa = smem + stage * NT_STAGE + wave * 128;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa), LDS_PTR(sa), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + a8), LDS_PTR(sa + 8 * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb), LDS_PTR(sa + NT_BK * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb + b8), LDS_PTR(sa + NT_BK * 128 + 8 * 128), 16, 0, 0);
    };
    issue(0, 0);
    __builtin_amdgcn_sched_barrier(0);      // the tile loads go out before the C loads (whose consumers wait vmcnt(0))

    // --- C tile -> accumulators: acc[a][b][r] = C(i0 + 64 wi + 32 (b>>1) + 2 (l4 + 4r) + (b&1), j0 + 32 wj + 2 l15 + a)
    v4d acc[2][4];
    double* cp[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        cp[a] = C + (size_t) (j0 + 32 * wj + 2 * l15 + a) * ldc + i0 + 64 * wi + 2 * l4;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v2d v = CEIL ? (v2d){1.0, 2.0} : *reinterpret_cast<const v2d*>(cp[a] + 32 * c + 8 * r);
                acc[a][2 * c][r] = v[0];
                acc[a][2 * c + 1][r] = v[1];
            }
    }
    __syncthreads();                                   // vmcnt(0): stage 0 and the C tile have landed
    if (STAMP) t1 = __builtin_amdgcn_s_memtime();

    const int nk = K / NT_BK;
    const int aoff = 64 * wi + 2 * l15, boff = NT_BK * 128 + 32 * wj + 2 * l15;
    // LDS reads run one 4-deep step ahead of the MFMAs that consume them (two fragment sets, order pinned with
    // sched_barrier: left alone, hipcc sinks every read to its first use and waits lgkmcnt(0) in front of each MFMA group)
    v2d fa[2][2], fb[2];
#define NT_READ(set, ks)                                                                   \
    do {                                                                                   \
        const double* row_ = st + (ks) * 4 * 128;                                          \
        fa[set][0] = *reinterpret_cast<const v2d*>(row_ + aoff);                           \
        fa[set][1] = *reinterpret_cast<const v2d*>(row_ + aoff + 32);                      \
        fb[set] = *reinterpret_cast<const v2d*>(row_ + boff);                              \
    } while (0)
#define NT_MMA(set)                                                                                                   \
    do {                                                                                                              \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                               \
            acc[a][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][0][0], fb[set][a], acc[a][0], 0, 0, NEG ? 1 : 0); \
            acc[a][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][0][1], fb[set][a], acc[a][1], 0, 0, NEG ? 1 : 0); \
            acc[a][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][1][0], fb[set][a], acc[a][2], 0, 0, NEG ? 1 : 0); \
            acc[a][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][1][1], fb[set][a], acc[a][3], 0, 0, NEG ? 1 : 0); \
        }                                                                                                             \
    } while (0)
    if constexpr (IL == 1) {
        // read `which` (0: fa[set][0], 1: fb[set], 2: fa[set][1] -- the order the step's first MFMAs need them) of step ks from stage st_
        auto rd = [&](int set, const double* st_, int ks, int which) {
            const double* row_ = st_ + ks * 4 * 128;
            if (which == 0) fa[set][0] = *reinterpret_cast<const v2d*>(row_ + aoff);
            else if (which == 1) fb[set] = *reinterpret_cast<const v2d*>(row_ + boff);
            else fa[set][1] = *reinterpret_cast<const v2d*>(row_ + aoff + 32);
        };
        auto mm = [&](int set, int j) {          // MFMA j of a step: a = j / 4, b = j % 4
            const int a = j >> 2, b = j & 3;
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][b >> 1][b & 1], fb[set][a], acc[a][b], 0, 0, NEG ? 1 : 0);
        };
        auto ld = [&](int kt_, int stage, int q) {      // tile load q of 4 (see issue())
            if (CEIL == 2) return;
            const double* pa = ga + (size_t) kt_ * NT_BK * lda;
            const double* pb = gb + (size_t) kt_ * NT_BK * ldbt;
            double* sa = smem + stage * NT_STAGE + wave * 128;
            if (q == 0) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa), LDS_PTR(sa), 16, 0, 0);
            else if (q == 1) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + a8), LDS_PTR(sa + 8 * 128), 16, 0, 0);
            else if (q == 2) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb), LDS_PTR(sa + NT_BK * 128), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb + b8), LDS_PTR(sa + NT_BK * 128 + 8 * 128), 16, 0, 0);
        };
#define NT_SB __builtin_amdgcn_sched_barrier(0)
        {
            const double* st0 = smem + l4 * 128;
            rd(0, st0, 0, 0); rd(0, st0, 0, 1); rd(0, st0, 0, 2);
        }
        NT_SB;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const double* st = smem + (kt & 1) * NT_STAGE + l4 * 128;
            const double* stn = smem + ((kt + 1) & 1) * NT_STAGE + l4 * 128;
            const int sn = (kt + 1) & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // step 0 (set 0): reads of step 1 and the next tile's four loads behind its MFMAs
                mm(0, j);
                if (j == 0) rd(1, st, 1, 0); else if (j == 2) rd(1, st, 1, 1); else if (j == 4) rd(1, st, 1, 2);
                else if (j == 1) ld(kt + 1, sn, 0); else if (j == 3) ld(kt + 1, sn, 1); else if (j == 5) ld(kt + 1, sn, 2);
                else if (j == 7) ld(kt + 1, sn, 3);
                NT_SB;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // step 1 (set 1)
                mm(1, j);
                if (j == 0) rd(0, st, 2, 0); else if (j == 2) rd(0, st, 2, 1); else if (j == 4) rd(0, st, 2, 2);
                NT_SB;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // step 2 (set 0)
                mm(0, j);
                if (j == 0) rd(1, st, 3, 0); else if (j == 2) rd(1, st, 3, 1); else if (j == 4) rd(1, st, 3, 2);
                NT_SB;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { mm(1, j); NT_SB; }     // step 3 (set 1), first half
            __syncthreads();                                     // all waves have read this stage; the next stage's loads have landed
            NT_SB;
#pragma unroll
            for (int j = 4; j < 8; ++j) {                        // second half, over the next tile's first fragment reads (set 0 is free)
                mm(1, j);
                if (j < 7) rd(0, stn, 0, j - 4);
                NT_SB;
            }
        }
        {   // last tile
            const double* st = smem + ((nk - 1) & 1) * NT_STAGE + l4 * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) { mm(0, j); if (j == 0) rd(1, st, 1, 0); else if (j == 2) rd(1, st, 1, 1); else if (j == 4) rd(1, st, 1, 2); NT_SB; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { mm(1, j); if (j == 0) rd(0, st, 2, 0); else if (j == 2) rd(0, st, 2, 1); else if (j == 4) rd(0, st, 2, 2); NT_SB; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { mm(0, j); if (j == 0) rd(1, st, 3, 0); else if (j == 2) rd(1, st, 3, 1); else if (j == 4) rd(1, st, 3, 2); NT_SB; }
#pragma unroll
            for (int j = 0; j < 8; ++j) mm(1, j);
        }
#undef NT_SB
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const double* st = smem + (kt & 1) * NT_STAGE + l4 * 128;
        NT_READ(0, 0);
        NT_READ(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        NT_MMA(0);
        __builtin_amdgcn_sched_barrier(0);
        NT_READ(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        NT_MMA(1);
        __builtin_amdgcn_sched_barrier(0);
        NT_READ(1, 3);
        __builtin_amdgcn_sched_barrier(0);
        NT_MMA(0);
        NT_MMA(1);
        __syncthreads();     // every wave is done with this stage; the next stage's loads (vmcnt(0)) have landed
    }
#undef NT_READ
#undef NT_MMA
    if (STAMP) t2 = __builtin_amdgcn_s_memtime();

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (CEIL) { if (acc[a][2 * c][r] == 0.12345678) *reinterpret_cast<v2d*>(cp[a] + 32 * c + 8 * r) = (v2d){acc[a][2 * c][r], acc[a][2 * c + 1][r]}; }
                else *reinterpret_cast<v2d*>(cp[a] + 32 * c + 8 * r) = (v2d){acc[a][2 * c][r], acc[a][2 * c + 1][r]};
            }
    if (STAMP && tid == 0) {
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        // 6 values per workgroup: the four stamps, where it ran (HW_ID | XCC_ID << 32) and its dispatch order
        unsigned long long* s = stamps + 6 * (size_t) blockIdx.x;
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));          // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));         // HW_REG_XCC_ID
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = (unsigned long long) hw | ((unsigned long long) (xcc & 0xf) << 32); s[5] = blockIdx.x;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5, the variant the round-4 review asked for: FOUR independent 4-wave workgroups per compute unit instead of two 8-wave ones.
// Same wave tile (64 x 32, 8 accumulators), same fragment maps, same occupancy (4 waves per SIMD); the workgroup tile is 128 x 64 and
// the k-tile 8 deep in THREE LDS stages of 12 KB (36 KB per workgroup), tiles requested two ahead and the barrier waiting only for the
// stage about to be read (s_waitcnt vmcnt(3)).  What it is after: a workgroup spends ~15 % of its time loading and storing its C tile,
// and with two workgroups per CU the matrix pipe runs on half its waves 42 % of the time (stamps of round 3); four workgroups with
// half-size tiles de-phase four ways.  What it costs: every tile of V serves 64 columns instead of 128 (x1.5 operand bytes L2 -> LDS).
// Measured (profiles/r05_nt_ceiling.txt): 60.8 against 59.0 TFLOP/s isolated, 51.2 against 49.8 in situ at 16384^2 (117.6 against 118.5 ms):
// the default since round 5; MI355XQR_NT4=0 restores the 8-wave kernel, =2 is a 3-per-CU form with 16-deep k-tiles (no gain).
// ------------------------------------------------------------------------------------------------
// RAG (round 6): M not a multiple of 128 (any even M) -- the bottom row tile is ragged.  Lanes neither load nor store the row pairs of C
// beyond M; the tile loader still brings 128 rows of A, so the caller guarantees that A's columns are readable up to the next multiple
// of 128 (qrd_gemm_nt4_ok: that fits A's leading dimension -- the rows are whatever the buffer holds and only reach accumulators that
// are never stored).  What it is for: outer blocks of 64 columns, where every other trailing matrix starts 64 rows into a tile and the
// update went to the generic kernels (4096^2 at nb 64: 0.20 against 0.15 ms per step), and matrices whose height is not a multiple of 128.
template <bool NEG, int CEIL, int BK4, int NS, bool RAG = false>
__global__ __launch_bounds__(256, (BK4 == 8 ? 4 : 3)) void gemm_nt4_kernel(int M, int N, int K, const double* __restrict__ A, int lda,
                                                          const double* __restrict__ Bt, int ldbt,
                                                          double* __restrict__ C, int ldc, int gx, int gy, int gm)
{
    // BK4 = 8, NS = 3: 36 KB of LDS, four workgroups per CU, tiles two ahead;  BK4 = 16, NS = 2: 48 KB, three per CU, tiles one ahead
    constexpr int STAGE = BK4 * 128 + BK4 * 64;           // doubles per stage: A image [BK4][128] + B image [BK4][64]
    constexpr int NLD = 3 * BK4 / 8;                       // tile-load instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    int tx, ty;
    tile_of_block(blockIdx.x, gx, gy, gm, tx, ty);
    const int i0 = tx * 128, j0 = ty * 64;
    // tile loader, instructions of 1 KiB: k-rows wave, wave + 4, .. of the A image; k-row pairs (2 wave, 2 wave + 1), (2 wave + 8, ..) of
    // the B image (32 lanes per 512-byte row)
    const double* ga = A + (size_t) wave * lda + i0 + 2 * lane;
    const double* gb = Bt + (size_t) (2 * wave + (lane >> 5)) * ldbt + j0 + 2 * (lane & 31);
    auto issue = [&](int kt, int stage) {
        if (CEIL == 2) return;
        const double* pa = ga + (size_t) kt * BK4 * lda;
        const double* pb = gb + (size_t) kt * BK4 * ldbt;
        double* sa = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < BK4 / 4; ++q)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + (size_t) (4 * q) * lda), LDS_PTR(sa + (wave + 4 * q) * 128), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < BK4 / 8; ++q)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb + (size_t) (8 * q) * ldbt), LDS_PTR(sa + BK4 * 128 + (wave + 4 * q) * 128), 16, 0, 0);
    };
    const int nk = K / BK4;
    issue(0, 0);
    if (NS == 3 && nk > 1) issue(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    v4d acc[2][4];
    double* cp[2];
    const int rlim = RAG ? M - (i0 + 64 * wi + 2 * l4) : 0;   // RAG: this lane's row pair 32 c + 8 r exists while 32 c + 8 r < rlim (M is even)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        cp[a] = C + (size_t) (j0 + 32 * wj + 2 * l15 + a) * ldc + i0 + 64 * wi + 2 * l4;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v2d v = (CEIL || (RAG && 32 * c + 8 * r >= rlim)) ? (v2d){1.0, 2.0} : *reinterpret_cast<const v2d*>(cp[a] + 32 * c + 8 * r);
                acc[a][2 * c][r] = v[0];
                acc[a][2 * c + 1][r] = v[1];
            }
    }
    __syncthreads();                                   // vmcnt(0): the stages requested so far and the C tile have landed
    const int aoff = 64 * wi + 2 * l15 + l4 * 128, boff = BK4 * 128 + 32 * wj + 2 * l15 + l4 * 64;
    constexpr int NKS = BK4 / 4;
    v2d fa[NKS][2], fb[NKS];
    if constexpr (BK4 == 8 && NS == 3) {
        // issue order spelled out as in gemm_nt_kernel<.., IL = 1>: one LDS read or tile load behind every other MFMA, the barrier in the
        // middle of the stage's second step with the next stage's first fragment reads behind it
        auto rd = [&](int set, const double* st_, int ks, int which) {
            if (which == 0) fa[set][0] = *reinterpret_cast<const v2d*>(st_ + ks * 4 * 128 + aoff);
            else if (which == 1) fb[set] = *reinterpret_cast<const v2d*>(st_ + ks * 4 * 64 + boff);
            else fa[set][1] = *reinterpret_cast<const v2d*>(st_ + ks * 4 * 128 + aoff + 32);
        };
        auto mm = [&](int set, int j) {
            const int a = j >> 2, b = j & 3;
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][b >> 1][b & 1], fb[set][a], acc[a][b], 0, 0, NEG ? 1 : 0);
        };
        auto ld = [&](int kt_, int q) {
            if (CEIL == 2) return;
            const double* pa = ga + (size_t) kt_ * BK4 * lda;
            const double* pb = gb + (size_t) kt_ * BK4 * ldbt;
            double* sa = smem + (kt_ % 3) * STAGE;
            if (q == 0) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa), LDS_PTR(sa + wave * 128), 16, 0, 0);
            else if (q == 1) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + (size_t) 4 * lda), LDS_PTR(sa + (wave + 4) * 128), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb), LDS_PTR(sa + BK4 * 128 + wave * 128), 16, 0, 0);
        };
#define N4_SB __builtin_amdgcn_sched_barrier(0)
        rd(0, smem, 0, 0); rd(0, smem, 0, 1); rd(0, smem, 0, 2);
        N4_SB;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const double* st = smem + (kt % 3) * STAGE;
            const double* stn = smem + ((kt + 1) % 3) * STAGE;
            const bool more = kt + 2 < nk;
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // step 0 (set 0): the reads of step 1 and the loads of the tile two ahead behind its MFMAs
                mm(0, j);
                if (j == 0) rd(1, st, 1, 0); else if (j == 2) rd(1, st, 1, 1); else if (j == 4) rd(1, st, 1, 2);
                else if (more && j == 1) ld(kt + 2, 0); else if (more && j == 3) ld(kt + 2, 1); else if (more && j == 5) ld(kt + 2, 2);
                N4_SB;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { mm(1, j); N4_SB; }     // step 1 (set 1), first half
            if (more) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            N4_SB;
#pragma unroll
            for (int j = 4; j < 8; ++j) {                        // second half, over the next stage's first fragment reads (set 0 is free)
                mm(1, j);
                if (j < 7) rd(0, stn, 0, j - 4);
                N4_SB;
            }
        }
        {   // last tile
            const double* st = smem + ((nk - 1) % 3) * STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) { mm(0, j); if (j == 0) rd(1, st, 1, 0); else if (j == 2) rd(1, st, 1, 1); else if (j == 4) rd(1, st, 1, 2); N4_SB; }
#pragma unroll
            for (int j = 0; j < 8; ++j) mm(1, j);
        }
#undef N4_SB
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt % NS;
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const double* st = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            fa[ks][0] = *reinterpret_cast<const v2d*>(st + ks * 4 * 128 + aoff);
            fa[ks][1] = *reinterpret_cast<const v2d*>(st + ks * 4 * 128 + aoff + 32);
            fb[ks] = *reinterpret_cast<const v2d*>(st + ks * 4 * 64 + boff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                acc[a][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][0][0], fb[ks][a], acc[a][0], 0, 0, NEG ? 1 : 0);
                acc[a][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][0][1], fb[ks][a], acc[a][1], 0, 0, NEG ? 1 : 0);
                acc[a][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][1][0], fb[ks][a], acc[a][2], 0, 0, NEG ? 1 : 0);
                acc[a][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][1][1], fb[ks][a], acc[a][3], 0, 0, NEG ? 1 : 0);
            }
        if (NS == 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (RAG && 32 * c + 8 * r >= rlim) continue;
                if (CEIL) { if (acc[a][2 * c][r] == 0.12345678) *reinterpret_cast<v2d*>(cp[a] + 32 * c + 8 * r) = (v2d){acc[a][2 * c][r], acc[a][2 * c + 1][r]}; }
                else *reinterpret_cast<v2d*>(cp[a] + 32 * c + 8 * r) = (v2d){acc[a][2 * c][r], acc[a][2 * c + 1][r]};
            }
}

// Measurement knobs (lab build only, qr_common.h): MI355XQR_NT4 = 1 (default): four workgroups per CU; 2: three per CU, k-tiles of 16;
// 0: the 8-wave kernel of round 3.  MI355XQR_NT_CEIL = 1 | 2: the ceiling experiment (profiles/r05_nt_ceiling.txt) -- C traffic / operand
// loads compiled out, RESULTS WRONG: those instantiations do not exist in the product library.
static int nt4(void)
{
    static const int v = QRD_LAB_ENV_INT("MI355XQR_NT4", 1);
    return v;
}

#ifdef QR_LAB
static int nt_ceil(void)
{
    static const int v = QRD_LAB_ENV_INT("MI355XQR_NT_CEIL", 0);
    return v;
}
#endif


static int nt_gm(void)
{
    static const int v = 8;     // read once, thread-safe
    return v;
}

// MI355XQR_NT_IL=0: the trailing update's K loop with grouped fragment reads (default 1: issue order spelled out, gemm_nt_kernel<.., 1>)
static int nt_il(void)
{
    static const int v = QRD_LAB_ENV_INT("MI355XQR_NT_IL", 1);
    return v;
}

static inline bool al16(const void* p, int ld) { return (((uintptr_t) p) & 15) == 0 && (ld & 1) == 0; }

extern "C" {

int qrd_gemm2_init(void)
{
    int rc = 0;
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT_STAGE * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT_STAGE * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 0, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 192 * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 0, 8, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 192 * 8);
#ifdef QR_LAB
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<true, 0, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 1, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 192 * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 2, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 192 * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 0, 16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * 192 * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 1, 16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * 192 * 8);
    rc |= (int) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<true, 2, 16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * 192 * 8);
#endif
    return rc;
}

// 1 if (M, N, K, operands) can go to gemm_nt_kernel as they are
int qrd_gemm_nt_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc)
{
    return M >= 128 && N >= 128 && M % 128 == 0 && N % 128 == 0 && K >= 16 && K % 16 == 0 && al16(A, lda) && al16(Bt, ldbt) && al16(C, ldc);
}

// 1 if C -= A Bt^T can go to the four-workgroup kernel (the trailing update's default): as above, or N a multiple of 64 only (its tiles are
// 128 x 64), or any even M with A readable to the next multiple of 128 rows (gemm_nt4_kernel<.., RAG>)
int qrd_gemm_nt4_ok(int M, int N, int K, const double* A, int lda, const double* Bt, int ldbt, const double* C, int ldc)
{
    if (nt4() != 1) return qrd_gemm_nt_ok(M, N, K, A, lda, Bt, ldbt, C, ldc);
    if (M < 128 || N < 64 || N % 64 || K < 32 || K % 16 || !al16(A, lda) || !al16(Bt, ldbt) || !al16(C, ldc)) return 0;
    if (M % 128 == 0) return 1;
    return M % 2 == 0 && (M + 127) / 128 * 128 <= lda;
}

// C -= A Bt^T (sign < 0) or C += A Bt^T (sign > 0) on the tile-aligned problem; gm < 0: library default
int qrd_gemm_nt(void* stream, int M, int N, int K, int sign, const double* A, int lda, const double* Bt, int ldbt,
                double* C, int ldc, int gm, unsigned long long* stamps)
{
    if (!stamps && sign < 0 && nt4() == 1 && !qrd_gemm_nt_ok(M, N, K, A, lda, Bt, ldbt, C, ldc) && qrd_gemm_nt4_ok(M, N, K, A, lda, Bt, ldbt, C, ldc)) {
        // N = 64 (mod 128) and / or M = 64 (mod 128): the four-workgroup kernel alone
        const int gx4 = (M + 127) / 128, gy4 = N / 64;
        if (gm < 0) gm = nt_gm();
        if (M % 128)
            hipLaunchKernelGGL((gemm_nt4_kernel<true, 0, 8, 3, true>), dim3(gx4 * gy4), dim3(256), 3 * (8 * 192) * sizeof(double), (hipStream_t) stream,
                               M, N, K, A, lda, Bt, ldbt, C, ldc, gx4, gy4, gm);
        else
            hipLaunchKernelGGL((gemm_nt4_kernel<true, 0, 8, 3>), dim3(gx4 * gy4), dim3(256), 3 * (8 * 192) * sizeof(double), (hipStream_t) stream,
                               M, N, K, A, lda, Bt, ldbt, C, ldc, gx4, gy4, gm);
        return (int) hipGetLastError();
    }
    if (!qrd_gemm_nt_ok(M, N, K, A, lda, Bt, ldbt, C, ldc)) return -7;
    const int gx = M / 128, gy = N / 128;
    if (gm < 0) gm = nt_gm();
    // (NT_SOLO, a measurement-only constant: ask for 100 KB of LDS, so that ONE workgroup fits a compute unit)
    static const int solo = 0;
    const size_t shm = solo ? (size_t) 100 * 1024 : 2 * NT_STAGE * sizeof(double);
    hipStream_t s = (hipStream_t) stream;
    if (!stamps && sign < 0 && nt4() && K >= 32) {
        // four workgroups per CU (k-tiles of 8, three stages)
        const int gy4 = N / 64;
        const dim3 g(gx * gy4), b(256);
#define NT4_LAUNCH(CE, BK, NS) hipLaunchKernelGGL((gemm_nt4_kernel<true, CE, BK, NS>), g, b, NS * (BK * 192) * sizeof(double), s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy4, gm)
#ifdef QR_LAB
        if (nt4() == 2) { if (nt_ceil() == 1) NT4_LAUNCH(1, 16, 2); else if (nt_ceil() == 2) NT4_LAUNCH(2, 16, 2); else NT4_LAUNCH(0, 16, 2); }
        else { if (nt_ceil() == 1) NT4_LAUNCH(1, 8, 3); else if (nt_ceil() == 2) NT4_LAUNCH(2, 8, 3); else NT4_LAUNCH(0, 8, 3); }
#else
        NT4_LAUNCH(0, 8, 3);
#endif
#undef NT4_LAUNCH
        return (int) hipGetLastError();
    }
#ifdef QR_LAB
    if (!stamps && sign < 0 && nt_ceil() == 1) {
        hipLaunchKernelGGL((gemm_nt_kernel<true, 0, 1, 1>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
        return (int) hipGetLastError();
    }
    if (!stamps && sign < 0 && nt_ceil() == 2) {
        hipLaunchKernelGGL((gemm_nt_kernel<true, 0, 1, 2>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
        return (int) hipGetLastError();
    }
#endif
    if (stamps)
        hipLaunchKernelGGL((gemm_nt_kernel<true, 1>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
    else if (sign < 0 && nt_il())
        hipLaunchKernelGGL((gemm_nt_kernel<true, 0, 1>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
    else if (sign < 0)
        hipLaunchKernelGGL((gemm_nt_kernel<true, 0>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<false, 0>), dim3(gx * gy), dim3(512), shm, s, M, N, K, A, lda, Bt, ldbt, C, ldc, gx, gy, gm, stamps);
    return (int) hipGetLastError();
}

}   // extern "C"
