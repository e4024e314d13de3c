// uvs_chol16.h -- Cholesky of ONE 16x16 diagonal block of the reduced system, the serial kernel of the blocked factorization
// (Ceres' dense/sparse Cholesky of the reduced camera matrix behind SPARSE_SCHUR, estimator.cpp:982-994).
//
// The 16 pivots of a block are a dependent chain; what matters is the latency of one link.  The block arrives in the FP64 MFMA
// C layout (row = (lane >> 4) + 4 reg, col = lane & 15) because the trailing update that produced it ran on the matrix cores.
// Here every 16-lane row of the wave gets a full copy of the matrix, ONE MATRIX ROW PER LANE (gfx950 v_permlane{16,32}_swap: no
// LDS round trip), and the factorization runs on the VALU in its right-looking form  a[r][c] -= L[r][j] L[c][j]  where L[c][j] -- lane c's
// register j -- reaches lane r as the DPP operand of the FMA itself (v_fmac_f64_dpp ... row_newbcast:c): one instruction per term, no
// readlane / SGPR round trip.  A link of the chain is  DPP FMA -> DPP broadcast of the pivot -> rsq + one third-order correction ->
// multiply  (measured ~95 cycles, tools/micro_dpp.hip: dependent DPP operation 21, rsq + correction 36, FP64 multiply 14) instead of
// ~250 for the readlane -> rcp -> rank-1 MFMA link it replaces.  The inverse W = L^-1 that the panel solve multiplies with is built on
// the side by the matrix core (one elementary elimination per pivot applied to the identity), off the chain.
#pragma once
#include <hip/hip_runtime.h>

namespace uvsdev {

typedef double d4c_t __attribute__((ext_vector_type(4)));

// value of lane J of this lane's 16-lane row (DPP row_newbcast, legal for 64-bit operands on gfx90a+).
// NOP = the source register was written by one of the two preceding VALU instructions: the VALU-write -> DPP-read hazard needs two
// wait states and the compiler cannot see through inline asm (an s_nop costs a full issue slot, ~8 cycles: only where required).
// volatile: the statements keep their program order, which is what the NOP placement relies on.
template <int J, bool NOP> __device__ __forceinline__ double row_bcast(double v) {
    double r;
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
// acc + a * row_bcast<J>(b)
template <int J, bool NOP> __device__ __forceinline__ double fmac_row_bcast(double acc, double a, double b) {
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
    else asm volatile("v_fmac_f64_dpp %0, %2, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
    return acc;
}

// every 16-lane row of the wave gets the values the four rows hold in `v`: out[s] = v of the lane with the same (lane & 15) in row s
__device__ __forceinline__ void rows_replicate(double v, double* out /*[4]*/) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    unsigned o[2][4];
    const unsigned w[2] = {lo, hi};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const auto h = __builtin_amdgcn_permlane32_swap(w[d], w[d], false, false);      // h[0] = rows {0,1,0,1}, h[1] = rows {2,3,2,3}
        const auto a = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);      // a[0] = row 0 everywhere, a[1] = row 1 everywhere
        const auto b = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
        o[d][0] = a[0]; o[d][1] = a[1]; o[d][2] = b[0]; o[d][3] = b[1];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) out[s] = __hiloint2double((int)o[1][s], (int)o[0][s]);
}

// Right-looking, one matrix row per lane:  after column j of L is known (register L[j] of every lane), row r takes
//     a[r][c] -= L[r][j] L[c][j],  c > j,      L[c][j] = lane c's L[j] = the DPP operand of the FMA.
// Only the update of column j + 1 sits on the pivot chain; the updates of the columns beyond it are DEFERRED into the latency
// shadows of the next pivot's reciprocal square root (independent accumulators, 8 cycles of issue each).
template <int JP, int C0, int C1> struct Chol16Defer {      // a[c] += nL[JP] * L[c][JP] for c in [C0, C1)
    static __device__ __forceinline__ void run(double* a, const double* L, const double* nL) {
        if constexpr (JP >= 0 && C0 < C1 && C0 < 16) { a[C0] = fmac_row_bcast<C0, false>(a[C0], nL[JP], L[JP]); Chol16Defer<JP, C0 + 1, C1>::run(a, L, nL); }
    }
};

template <int J, bool WITH_W> struct Chol16Step {
    // pivots J .. 15.  a: this lane's matrix row; L / nL: row of the factor and its negative; inv[j] = 1 / L[j][j] (same in every lane);
    // T: the elimination of the identity in C layout (W = L^-1 up to the row scaling by inv, applied by the caller)
    static __device__ __forceinline__ void run(double* a, double* L, double* nL, double* inv, d4c_t& T, int li, int lk, bool& ok) {
        if constexpr (J < 16) {
            // deferred updates of pivot J - 1 cover columns J + 1 .. 15, in three groups between the links of the reciprocal square root
            constexpr int D0 = J + 1, DN = 16 - D0, D1 = D0 + (DN + 2) / 3, D2 = D1 + (DN + 1) / 3;
            const double dsum = row_bcast<J, true>(a[J]);
            ok = ok && (dsum > 0.0);
            // 1/sqrt: hardware seed (v_rsq_f64, ~2^-26) and ONE third-order step  y (1 + e/2 + 3 e^2/8), e = 1 - x y^2  (~2^-75)
            const double y = __builtin_amdgcn_rsq(dsum);
            Chol16Defer<J - 1, D0, D1>::run(a, L, nL);
            const double e = fma(-dsum * y, y, 1.0);
            Chol16Defer<J - 1, D1, D2>::run(a, L, nL);
            const double iv = fma(y, e * fma(0.375, e, 0.5), y);
            Chol16Defer<J - 1, D2, 16>::run(a, L, nL);
            inv[J] = iv;
            L[J] = a[J] * iv; nL[J] = a[J] * -iv;
            if constexpr (J < 15) a[J + 1] = fmac_row_bcast<J + 1, true>(a[J + 1], nL[J], L[J]);      // the link of the chain
            if constexpr (J < 15 && WITH_W) {
                // elementary elimination of pivot J applied to T:  T[i][:] -= (L[i][J] / L[J][J]) T[J][:], i > J.
                // A operand: A[i][k] at lane i + 16 k, non-zero only in k-slot J & 3; B operand: B[k][c] at lane c + 16 k = row 4 (J >> 2) + k of T.
                const double us = (lk == (J & 3) && li > J) ? nL[J] * iv : 0.0;
                T = __builtin_amdgcn_mfma_f64_16x16x4f64(us, T[J >> 2], T, 0, 0, 0);
            }
            Chol16Step<J + 1, WITH_W>::run(a, L, nL, inv, T, li, lk, ok);
        }
    }
};

// Factor the symmetric positive definite 16x16 block held in C layout in `acc` (both triangles valid).
//   Dk   : the block's LDS storage, row stride `ld`:  lower triangle <- L,  strictly upper (m, j), m < j  <- W[j][m]  (W = L^-1)
//   dinv : 16 doubles <- 1 / L[j][j]
// Returns false when a pivot is not positive (the caller raises its failure flag).  One wavefront, all 64 lanes.
template <bool WITH_W = true>
__device__ __forceinline__ bool chol16_factor(const d4c_t& acc, int lane, double* Dk, int ld, double* dinv) {
    const int li = lane & 15, lk = lane >> 4;
    double a[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // acc[q] of lane (li, s) = A[s + 4 q][li] = A[li][s + 4 q]
        double r4[4];
        rows_replicate(acc[q], r4);
#pragma unroll
        for (int s = 0; s < 4; ++s) a[4 * q + s] = r4[s];
    }
    double L[16], nL[16], inv[16];
    d4c_t T;
#pragma unroll
    for (int q = 0; q < 4; ++q) T[q] = (lk + 4 * q == li) ? 1.0 : 0.0;
    bool ok = true;
    Chol16Step<0, WITH_W>::run(a, L, nL, inv, T, li, lk, ok);
    // stores: register q of this lane belongs to column j = lk + 4 q of row li
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lk + 4 * q;
        const double Lj = lk == 0 ? L[4 * q] : lk == 1 ? L[4 * q + 1] : lk == 2 ? L[4 * q + 2] : L[4 * q + 3];
        const double ij = lk == 0 ? inv[4 * q] : lk == 1 ? inv[4 * q + 1] : lk == 2 ? inv[4 * q + 2] : inv[4 * q + 3];
        // T[q] of this lane = T[row j][col li]:  W[j][li] = T[j][li] / L[j][j], stored transposed at (li, j) of the upper triangle
        Dk[li * ld + j] = (li >= j) ? Lj : T[q] * ij;
        if (li == j) dinv[j] = ij;
    }
    return ok;
}

}  // namespace uvsdev
