// K1 (fast path)  lqr_backward, fp64, LDS-DMA staged, 12 MFMA / step -- gfx950.
//
// Arithmetic: the reference's Joseph-form recursion (zopt/lqrUtils.py:167-172), per trajectory
//     V <- Q[T-1];  for k = T-1..0:  L_k = solve(R_k + B_k^T V B_k, B_k^T V A_k)
//                                    V   = Q_k + L_k^T R_k L_k + (A_k - B_k L_k)^T V (A_k - B_k L_k)
// on the tile-16 mapping of tile16_f64.h (one wave64 per trajectory, stacked index [x | u], n + m = 16 or 12).
//
// What the hardware measurements dictated (profiles/README.md): on gfx950 the fp64 MFMA (64 cycles) and the
// fp64/32-bit VALU share one issue pipe (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so a step costs
// 64 * #MFMA + 4 * #VALU cycles per wave and four waves per SIMD add up.  Hence:
//
//   operands   HBM --global_load_lds_dwordx4 (16 B/lane, coalesced, memory order)--> LDS ring slot --ds_read_b64-->
//              MFMA register layouts.  In-flight data costs no VGPRs, the ring runs D = 3 steps ahead, B^T / R
//              layouts are just other LDS reads of the same bytes; lanes that must read 0 read the slot's padding,
//              which idle DMA lanes fill from a zero source (no per-step masking instructions).
//   Y  = V^T F                    3 MFMA     F = [A_k | B_k]
//   G  = Y^T F + [0 ; R]          3 MFMA     rows n.. of G = [B^T V A | R + B^T V B]
//   L  = solve(Suu, Sux)          4 x 16 tile through LDS; lane-local elimination WITHOUT row exchanges when every
//                                 multiplier is <= 4 in magnitude (checked, wave-uniform), else the pivoted LU
//   [Acl ; -R L] = [A ; 0] + [B ; R] (-L)     1 MFMA   (stacked A operand: rows < n give A - B L, rows n.. give -R L)
//   W  = V^T Acl = Y_A + Y_B (-L) 1 MFMA     (Y_B = V^T B is already in Y; transposed through LDS off the critical path)
//   V' = Q + (-L)^T (-R L) + W^T Acl         4 MFMA
//
// Requirements: n in {8, 12}, m = 4 (every per-step matrix a multiple of 16 B, all K-step rows live) and 16-B aligned
// base pointers; every other supported shape runs the register-prefetch kernel in lqr_backward.hip.
#include "dma_ring.h"
#include "tile16_f64.h"
#include "zm_common.h"

#include <cstdlib>

namespace zm {

// ES: bytes per stored element (8: fp64 arrays; 4: fp32 arrays -- the arithmetic is fp64 either way, operands are widened as they
// are read from the ring and L_k is narrowed as it is stored: half the HBM bytes for fp32 callers, no conversion passes)
template <int N, int M, int ES = 8>
struct DmaGeom {
    static constexpr int KS = N / 4;
    static constexpr int NP = N;
    static constexpr int EPC = 16 / ES;                                          // elements per 16-B chunk
    static constexpr int CA = N * N / EPC, CB = N * M / EPC, CR = M * M / EPC;   // 16-B chunks per step
    static constexpr int CT = 2 * CA + CB + CR;
    static constexpr int NI = (CT + 63) / 64;   // DMA wave-instructions per step
    static constexpr int SLOT = NI * 1024;      // bytes per ring slot
    static constexpr int OA = 0, OB = CA * 16, OQ = (CA + CB) * 16, OR = (2 * CA + CB) * 16;  // byte offsets in a slot
    static constexpr int OZ = CT * 16;          // first byte of the zero padding
    static_assert(N % 4 == 0 && N >= 4 && N <= 12 && M == 4, "fast path: all K-step rows live, m = 4");
    static_assert(SLOT - OZ >= 16, "slot needs zero padding");
};

template <int N, int M, int D, int ES = 8>
struct DmaState {
    using G = DmaGeom<N, M, ES>;
    const char* p[G::NI];  // per-lane source address of the next step to fetch, one per DMA instruction
    int st[G::NI];         // per-lane byte stride between consecutive steps of that array (0 for the zero source)
    int oF, dF;            // LDS byte offset of F[g][c] in a slot and its K-step stride (A lanes 4n*8, B lanes 4m*8)
    int oQ, oRm, oBR;      // LDS byte offsets (in a slot) of Q[g][c], R[g][c-n] (or zero pad), [B ; R][c][g]
};

// AUX: cache-policy bits of the DMA (sc0 = 1, nt = 2, sc1 = 16).  The product streams its inputs non-temporally (2): every byte
// is read exactly once, and keeping it out of the L2 / Infinity Cache replacement order is worth 10 % of the kernel's time
// (tools/k1_lab.hip: 144 -> 128 us per launch; the memory-only variant of the same access pattern 134 -> 119 us).
template <int N, int M, int D, int AUX = 2, int ES = 8>
__device__ __forceinline__ void dma_issue(DmaState<N, M, D, ES>& a, char* slot) {
    using G = DmaGeom<N, M, ES>;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)a.p[i], (lds_void_t*)(slot + i * 1024), 16, 0, AUX);
        a.p[i] -= a.st[i];
    }
}

// X: diagnostic bits for tools/k1_lab.hip (0 in the product): 1 = every block loads the inputs of trajectory blockIdx % 64 (the
// working set then sits in L2: compute time without HBM latency / bandwidth), 2 = no s_setprio around the solve,
// 4 = s_memtime stamps per segment, summed into zm_k1_stamps (diagnostic build only: the stamps' waits forbid overlaps).
#ifdef ZM_K1_LAB
__device__ unsigned long long zm_k1_stamps[8];
#define ZM_STAMP(var)                                                                          \
    if constexpr ((X & 4) != 0) {                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#else
#define ZM_STAMP(var)
#endif

// W: waves per workgroup.  Wave w of block b owns trajectory b * W + w; the waves never synchronise or share data -- a larger
// workgroup only makes the co-resident waves of a CU work on ADJACENT trajectories (fewer distinct pages per CU).
// LDS write -> read by another lane of the wave: zm::wave_lds_sync (zm_common.h: compiler-level barrier); HEAVY = the round-1 form
// (fences + wave barrier), kept for the lab's A/B (X bit 21).
template <bool HEAVY>
__device__ __forceinline__ void k1_lds_sync() {
    if constexpr (HEAVY) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        wave_lds_sync();
    }
}

// one stored element of the ring, widened to the arithmetic type
template <typename IO>
__device__ __forceinline__ double ring_read(const char* p) {
    return (double)*(const IO*)p;
}

template <int N, int M, int D, bool G4, int X = 0, int W = 1, int WPS = 4, typename IO = double>
__global__ __launch_bounds__(64 * W, WPS) void lqr_backward_dma_f64(const IO* __restrict__ A,
                                                                  const IO* __restrict__ B,
                                                                  const IO* __restrict__ Q,
                                                                  const IO* __restrict__ R, IO* __restrict__ L,
                                                                  const int T, const long batch) {
    constexpr int ES = (int)sizeof(IO);
    static_assert(ES == 8 || X == 0, "the lab's diagnostic variants are fp64-storage only");
    using G = DmaGeom<N, M, ES>;
    constexpr int KS = G::KS, NI = G::NI, SLOT = G::SLOT;
    constexpr int kAux = (X & 16) ? ((X >> 8) & 31) : 2;   // lab: bit 16 selects the policy in bits 8..12; product: nt
    constexpr int nn = N * N, nm = N * M, mm = M * M;
    // ONE LDS object: D ring slots | 4x16 exchange tile of the solve | Y_B = V^T B (n x 4) for the W product.
    constexpr int EXCH = D * SLOT, YBO = EXCH + 64 * 8;
    constexpr int PER_WAVE = YBO + N * M * 8;
    __shared__ __attribute__((aligned(16))) char lds_all[W * PER_WAVE];
    const int wave = W == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char* const lds = lds_all + wave * PER_WAVE;
    double* exch = (double*)(lds + EXCH);

    const int lane = threadIdx.x & 63;
    const long traj = (long)blockIdx.x * W + wave;
    if (traj >= batch) return;   // (whole waves only: no barrier anywhere in this kernel)
    const long ltraj = (X & 1) ? (traj & 63) : traj;   // trajectory whose inputs are loaded (diagnostic bit 1)
    const int g = lane >> 4, c = lane & 15;
    const bool cA = c < N;                    // state column
    const bool cB = (c >= N) && (c < N + M);  // control column

    DmaState<N, M, D, ES> a;
    {
        const long last = ltraj * T + (T - 1);
        const char* At = (const char*)(A + last * nn);
        const char* Bt = (const char*)(B + last * nm);
        const char* Qt = (const char*)(Q + last * nn);
        const char* Rt = (const char*)(R + last * mm);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = i * 64 + lane;  // 16-B chunk index inside the step's [A|B|Q|R|0-pad] image
            if (q < G::CA) {
                a.p[i] = At + q * 16;
                a.st[i] = nn * ES;
            } else if (q < G::CA + G::CB) {
                a.p[i] = Bt + (q - G::CA) * 16;
                a.st[i] = nm * ES;
            } else if (q < 2 * G::CA + G::CB) {
                a.p[i] = Qt + (q - G::CA - G::CB) * 16;
                a.st[i] = nn * ES;
            } else if (q < G::CT) {
                a.p[i] = Rt + (q - 2 * G::CA - G::CB) * 16;
                a.st[i] = mm * ES;
            } else {  // idle lanes: zero-fill the slot's padding
                a.p[i] = (const char*)zm_zero_src;
                a.st[i] = 0;
            }
        }
    }
    // MFMA operand addresses inside a slot.  Lanes outside a matrix either read finite don't-care data (they only
    // ever feed padding rows/columns of the tile) or, where a true zero is required (R's accumulator init under
    // the state columns), the zero padding.
    a.oF = cA ? (G::OA + (g * N + c) * ES) : cB ? (G::OB + (g * M + (c - N)) * ES) : G::OZ;
    a.dF = cA ? 4 * N * ES : cB ? 4 * M * ES : 0;
    a.oQ = G::OQ + (g * N + (cA ? c : 0)) * ES;
    a.oRm = cB ? (G::OR + (g * M + (c - N)) * ES) : G::OZ;
    a.oBR = cA ? (G::OB + (c * M + g) * ES) : cB ? (G::OR + ((c - N) * M + g) * ES) : G::OZ;
    const int oYBw = YBO + (g * M + (c - N)) * 8;           // Y_B[4r+g][c-n]   (+ r*4*M*8), lanes cB
    const int oYBr = YBO + ((cA ? c : 0) * M + g) * 8;      // Y_B[c][g]        A operand of Y_B (-L)
    const int oYBa = YBO + (g * M + (c & 3)) * 8;           // Y_B[4s+g][c & 3] (+ s*4*M*8): A operand of the 4x4x4 blocks (G4)
    const bool vL = cA;                                     // (g < M always: M == 4)
    IO* pL = L + ((((X >> 19) & 1) ? (traj & 63) : traj) * T + (T - 1)) * nm + g * N + c;

    // prologue: fill the ring with steps T-1 .. T-D
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (((X >> 20) & 1) == 0 && T - 1 - i >= 0) dma_issue<N, M, D, kAux, ES>(a, lds + i * SLOT);

    double V[KS];
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, acc[5] = {0, 0, 0, 0, 0};
    (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)st5; (void)acc;
    int j = T - 1;  // step index; step j lives in slot (T-1-j) % D
    bool first = true;
    for (;;) {
#pragma unroll
        for (int si = 0; si < D; ++si) {
            char* slot = lds + si * SLOT;
            ZM_STAMP(st0)
            wait_for_step<NI, D, (X & 8) != 0>(j, T);
            d4 f4 = zero4(), q4 = zero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) f4[s] = ring_read<IO>(slot + a.oF + s * a.dF);
#pragma unroll
            for (int s = 0; s < KS; ++s) q4[s] = ring_read<IO>(slot + a.oQ + s * (4 * N * ES));
            const double rm = ring_read<IO>(slot + a.oRm);
            const double br = ring_read<IO>(slot + a.oBR);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // operands are in registers: the slot may be refilled
            if (((X >> 20) & 1) == 0 && j - D >= 0) dma_issue<N, M, D, kAux, ES>(a, slot);
            ZM_STAMP(st1)
#ifdef ZM_K1_LAB
            if constexpr ((X & 32) != 0) {   // memory only: the access pattern's own time (no MFMA, no solve)
                const double lvm = f4[0] + f4[1] + f4[2] + q4[0] + q4[1] + q4[2] + rm + br;
                constexpr int SM = (X >> 13) & 7;   // store shape (diagnostic): 0 = product (48 lanes x 8 B per step)
                typedef double d2 __attribute__((ext_vector_type(2)));
                double* const Lk = L + (traj * T + j) * nm;     // L_j: 384 B
                if constexpr ((X & 128) != 0) { if (lvm == 123.456 && vL) *pL = lvm; }
                else if constexpr (SM == 1) { if (lane < 24) *(d2*)(Lk + 2 * lane) = d2{lvm, lvm}; }                 // 24 lanes x 16 B per step
                else if constexpr (SM == 2) { if ((j & 1) == 0 && lane < 48 && j + 1 < T) *(d2*)(Lk + 2 * lane) = d2{lvm, lvm}; }   // 48 x 16 B per 2 steps
                else if constexpr (SM == 3) { if ((j & 3) == 0 && j + 3 < T) { *(d2*)(Lk + 2 * lane) = d2{lvm, lvm}; if (lane < 32) *(d2*)(Lk + 128 + 2 * lane) = d2{lvm, lvm}; } }
                else if constexpr (SM == 4) { if ((j & 7) == 0 && j + 7 < T) { *(d2*)(Lk + 2 * lane) = d2{lvm, lvm}; *(d2*)(Lk + 128 + 2 * lane) = d2{lvm, lvm}; *(d2*)(Lk + 256 + 2 * lane) = d2{lvm, lvm}; } }
                else if (vL) { if constexpr ((X & 64) != 0) __builtin_nontemporal_store(lvm, pL); else *pL = lvm; }
                pL -= nm;
                if (--j < 0) return;
                continue;
            }
#endif
            if (first) {
#pragma unroll
                for (int s = 0; s < KS; ++s) V[s] = q4[s];  // V <- Q[T-1]   (lqrUtils.py:172)
                first = false;
            }

            // Y = V^T F
            d4 y = zero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) y = mfma(V[s], f4[s], y);
            // Y_B = Y[:, n..n+3] -> LDS (read back transposed for W, and -- G4 -- as the small A operand below)
            if (cB) {
#pragma unroll
                for (int s = 0; s < KS; ++s) *(double*)(lds + oYBw + s * (4 * M * 8)) = y[s];
            }
            double srow;
            if constexpr (G4) {
                // [Sux | Suu] = Y_B^T F + [0 | R]: only 4 of the 16 rows of F^T V F are needed, so instead of 3 full-tile MFMAs
                // (64 cycles each) the 4 x 16 result is computed as four 4x4 blocks by v_mfma_f64_4x4x4_4b (16 cycles each):
                // block q = c >> 2 takes B_q[k][j] = F[4s+k][4q+j] -- exactly the registers f4[s] already hold -- and
                // A_q[u][k] = Y_B[4s+k][u], read back from LDS as Y_B[4s+g][c & 3]; lane (g, c) receives S[g][c].
                k1_lds_sync<((X >> 21) & 1) != 0>();
                double ya[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) ya[s] = *(const double*)(lds + oYBa + s * (4 * M * 8));
                srow = rm;   // + R under the control columns, + 0.0 from the padding elsewhere
#pragma unroll
                for (int s = 0; s < KS; ++s) srow = __builtin_amdgcn_mfma_f64_4x4x4f64(ya[s], f4[s], srow, 0, 0, 0);
            } else {
                // G = Y^T F + [0 ; R]  -> row n+g : [ B^T V A | R + B^T V B ]
                d4 gacc = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) gacc = mfma(y[s], f4[s], gacc);
                srow = gacc[KS] + rm;  // (+ R under the control columns, + 0.0 from the padding elsewhere)
            }

            ZM_STAMP(st2)
            // m x m solve: 4 x 16 tile through LDS, every lane reads Suu (broadcast) and its own RHS column
            exch[g * 16 + c] = srow;
            k1_lds_sync<((X >> 21) & 1) != 0>();
            double S[4][4], b[4], x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) S[i][jj] = exch[i * 16 + N + jj];
                b[i] = exch[i * 16 + c];
            }
            const double ybt = *(const double*)(lds + oYBr);
            ZM_STAMP(st3)
            // The solve is a long chain of short dependent VALU ops; without priority each of them can queue behind a
            // 64-cycle MFMA of another wave on the shared fp64 pipe (measured: -3..4 % kernel time).
            if constexpr ((X & 2) == 0) __builtin_amdgcn_s_setprio(3);
            // wave-uniform vote on the scalar unit: "some lane failed the growth check" = ballot(!ok) != 0
            if (__builtin_amdgcn_ballot_w64(!lu_solve4_nopivot(S, b, x)) != 0ull) {
                // rare: growth check failed somewhere in the wave -> partial pivoting, IEEE division
                lu_solve4_fallback(S, b, x);
            }
            if constexpr (((X >> 21) & 1) != 0) __builtin_amdgcn_wave_barrier();
            const double x01 = (g & 1) ? x[1] : x[0];
            const double x23 = (g & 1) ? x[3] : x[2];
            const double lv = (g & 2) ? x23 : x01;  // L_k[g][c]
            const double ln = -lv;
            if constexpr ((X & 2) == 0) __builtin_amdgcn_s_setprio(0);
            if (vL) {
#ifdef ZM_K1_LAB
                constexpr int SP = (X >> 16) & 7;   // store cache policy (diagnostic): 1 = sc1, 2 = sc0 sc1, 3 = nt sc1, 4 = nt sc0 sc1
                if constexpr (SP == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(pL), "v"(lv) : "memory");
                else if constexpr (SP == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(pL), "v"(lv) : "memory");
                else if constexpr (SP == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(pL), "v"(lv) : "memory");
                else if constexpr (SP == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(pL), "v"(lv) : "memory");
                else
#endif
                if constexpr ((X & 64) != 0) __builtin_nontemporal_store((IO)lv, pL);
                else *pL = (IO)lv;
            }
            pL -= nm;
            ZM_STAMP(st4)

            // [Acl ; -R L] = [A ; 0] + [B ; R] (-L)
            f4 = mfma(br, ln, f4);
            // W = Y + Y_B (-L)  = V^T (A - B L)
            y = mfma(ybt, ln, y);
            // V' = Q + (-L)^T (-R L) + W^T Acl
            q4 = mfma(ln, f4[KS], q4);
#pragma unroll
            for (int s = 0; s < KS; ++s) q4 = mfma(y[s], f4[s], q4);
#pragma unroll
            for (int s = 0; s < KS; ++s) V[s] = q4[s];
#ifdef ZM_K1_LAB
            if constexpr ((X & 4) != 0) {
                asm volatile("" ::"v"(V[0]), "v"(V[KS - 1]));
                ZM_STAMP(st5)
                acc[0] += st1 - st0; acc[1] += st2 - st1; acc[2] += st3 - st2; acc[3] += st4 - st3; acc[4] += st5 - st4;
                if (j == 0 && lane == 0) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) atomicAdd(&zm_k1_stamps[q], acc[q]);
                    atomicAdd(&zm_k1_stamps[5], 1ull);
                }
            }
#endif
            if (--j < 0) return;
        }
    }
}

template <int N, int M>
static int launch_dma(const double* A, const double* B, const double* Q, const double* R, double* L, int64_t batch,
                      int T, hipStream_t stream) {
    // default: the [Sux | Suu] rows by v_mfma_f64_4x4x4_4b blocks (+2.6 % at steady state); ZOPT_AMD_LQR_G4=0 selects the three
    // full-tile MFMAs instead (A/B measurements)
    static const bool g4 = [] {
        const char* e = zm::lab_env("ZOPT_AMD_LQR_G4");
        return !(e && e[0] == '0');
    }();
    static const int depth = [] {   // ring depth (steps in flight per wave): 3 unless ZOPT_AMD_LQR_D=2 (A/B measurements)
        const char* e = zm::lab_env("ZOPT_AMD_LQR_D");
        return (e && e[0] == '2') ? 2 : 3;
    }();
    const dim3 grid((unsigned)batch), block(64);
    if (!g4)
        hipLaunchKernelGGL((lqr_backward_dma_f64<N, M, 3, false>), grid, block, 0, stream, A, B, Q, R, L, T, (long)batch);
    else if (depth == 2)
        hipLaunchKernelGGL((lqr_backward_dma_f64<N, M, 2, true>), grid, block, 0, stream, A, B, Q, R, L, T, (long)batch);
    else
        hipLaunchKernelGGL((lqr_backward_dma_f64<N, M, 3, true>), grid, block, 0, stream, A, B, Q, R, L, T, (long)batch);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}

// fp32 arrays in and out, fp64 arithmetic (the same kernel instantiated on 4-byte ring elements)
template <int N, int M>
static int launch_dma_f32io(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch, int T,
                            hipStream_t stream) {
    hipLaunchKernelGGL((lqr_backward_dma_f64<N, M, 3, true, 0, 1, 4, float>), dim3((unsigned)batch), dim3(64), 0, stream, A, B, Q, R, L,
                       T, (long)batch);
    ZM_HIP_CHECK(hipGetLastError());
    return ZM_OK;
}
int lqr_backward_dma_dispatch_f32io(const float* A, const float* B, const float* Q, const float* R, float* L, int64_t batch, int T,
                                    int n, int m, hipStream_t stream) {
    const uintptr_t al = (uintptr_t)A | (uintptr_t)B | (uintptr_t)Q | (uintptr_t)R;
    if (al & 15) return ZM_EUNSUPPORTED;
    if (n == 12 && m == 4) return launch_dma_f32io<12, 4>(A, B, Q, R, L, batch, T, stream);
    if (n == 8 && m == 4) return launch_dma_f32io<8, 4>(A, B, Q, R, L, batch, T, stream);
    return ZM_EUNSUPPORTED;
}

// Returns ZM_EUNSUPPORTED when the shape / alignment is not covered so that the caller falls back.
int lqr_backward_dma_dispatch(const double* A, const double* B, const double* Q, const double* R, double* L,
                              int64_t batch, int T, int n, int m, hipStream_t stream) {
    const uintptr_t al = (uintptr_t)A | (uintptr_t)B | (uintptr_t)Q | (uintptr_t)R;
    if (al & 15) return ZM_EUNSUPPORTED;
    if (n == 12 && m == 4) return launch_dma<12, 4>(A, B, Q, R, L, batch, T, stream);
    if (n == 8 && m == 4) return launch_dma<8, 4>(A, B, Q, R, L, batch, T, stream);
    return ZM_EUNSUPPORTED;
}

}  // namespace zm
