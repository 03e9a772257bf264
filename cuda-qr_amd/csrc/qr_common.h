// qr_common.h -- device helpers shared by the HIP translation units
#ifndef QR_COMMON_H
#define QR_COMMON_H
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define LEAFW 32

// Measurement knobs (settled A/Bs, ceiling experiments) are environment variables only in the LAB build (`make lab`: libmi355xqr_lab.so,
// -DQR_LAB); in the product library they are the constants given here, so that no stray variable changes what a launch does.
#ifdef QR_LAB
#include <stdlib.h>
#define QRD_LAB_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define QRD_LAB_ENV_INT(name, dflt) (dflt)
#endif

// ---- cross-lane moves on the VALU (no LDS traffic) -----------------------------------------------------
// ds_bpermute-based shuffles go through the CU's single LDS pipeline; with 8 waves of a workgroup reducing 32
// sums per Householder column that pipeline, not the SIMDs, set the pace (measured 3 us per column).  gfx950
// has v_permlane32_swap / v_permlane16_swap and DPP row modes, which run on the SIMD that issues them.
__device__ __forceinline__ double dpp_f64(double v, int which)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (which) {       // `which` is always a literal after inlining
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;   // row_mirror      (lane ^ 15)
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;   // row_half_mirror (lane ^ 7)
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;     // quad_perm 2301  (lane ^ 2)
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;    // quad_perm 1032  (lane ^ 1)
    }
    return __hiloint2double(hi, lo);
}

// a (kept by the lower half / even rows), b (kept by the upper half / odd rows): returns, in every lane, the
// sum over the lane pair of the value that lane keeps.  SWAP32: pairs (l, l^32); else pairs (l, l^16).
template <bool SWAP32>
__device__ __forceinline__ double swap_add(double a, double b)
{
    const unsigned alo = (unsigned) __double2loint(a), ahi = (unsigned) __double2hiint(a);
    const unsigned blo = (unsigned) __double2loint(b), bhi = (unsigned) __double2hiint(b);
    if (SWAP32) {
        auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int) rh[0], (int) rl[0]) + __hiloint2double((int) rh[1], (int) rl[1]);
    } else {
        auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int) rh[0], (int) rl[0]) + __hiloint2double((int) rh[1], (int) rl[1]);
    }
}

// Transposing wave butterfly.  in: a[c], c < 32, per lane.  out (every lane): the sum over the 64 lanes of
// a[(lane >> 1) & 31] -- 32 cross-lane exchanges for 32 sums instead of 32 x 6.  Stage pairings: lane^32, lane^16
// (permlane swaps), lane^15, lane^7 (DPP row mirrors), lane^2, lane^1 (DPP quad perms); the masks span GF(2)^6,
// so after the six stages every lane holds a full 64-lane sum.  Selector bits 5,4,3,2,1 pick which half a lane keeps.
__device__ __forceinline__ double wave_reduce32(const double (&a)[LEAFW], int lane)
{
    double b16[16], b8[8], b4[4], b2[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) b16[q] = swap_add<true>(a[q], a[q + 16]);
#pragma unroll
    for (int q = 0; q < 8; ++q) b8[q] = swap_add<false>(b16[q], b16[q + 8]);
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double send = up ? b8[q] : b8[q + 4], keep = up ? b8[q + 4] : b8[q];
            b4[q] = keep + dpp_f64(send, 0);
        }
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double send = up ? b4[q] : b4[q + 2], keep = up ? b4[q + 2] : b4[q];
            b2[q] = keep + dpp_f64(send, 1);
        }
    }
    const bool up = (lane & 2) != 0;
    double v = (up ? b2[1] : b2[0]) + dpp_f64(up ? b2[0] : b2[1], 2);
    v += dpp_f64(v, 3);
    return v;
}
// Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2), so the
// linear id is first turned into "XCD-major" order (each XCD gets a contiguous range of tile indices), and tile
// indices walk the grid in groups of GM row tiles x all column tiles: the ~64 workgroups an XCD runs at a time then
// form a compact GM x (64/GM) block of the tile grid that shares GM V tiles and 64/GM Wt tiles in that XCD's L2.
// Speed only: any bijection is correct.
__device__ __forceinline__ void tile_of_block(int bid, int gx, int gy, int gm, int& tx, int& ty)
{
    const int nwg = gx * gy;
    int id = bid;
    if (gm > 0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int per = gm * gy, g = id / per, in = id - g * per;
        const int rows = min(gm, gx - g * gm);            // the last group may be shorter
        tx = g * gm + in % rows;
        ty = in / rows;
    } else {
        tx = id % gx;
        ty = id / gx;
    }
}

#endif
