// ftmpc_solve.hip -- kernel 2 of the MPC QP-step path: condensed-QP build + primal-dual IPM (fp32).
//
// ONE WAVEFRONT (64 lanes, one workgroup) PER QP INSTANCE, persistent over the batch
// (instance = blockIdx.x, += gridDim.x).  Everything is kept in the MFMA ACCUMULATOR LAYOUT of a
// 16x16 tile (lane (q, col), register s = element (4q+s, col)): register s of row-group q' is then
// contraction index k = 4q'+s, so BOTH operands of X'Y are plain accumulator registers (mm_tn).
// For its instance the wave
//   1. propagates the horizon-stacked input-to-state map  G_{k+1} = A_k G_k | B_k D_act  as NB
//      accumulator tiles with v_mfma_f32_16x16x4_f32 against dense LDS images of the stage matrices
//      (the stage records come from ftmpc_linearize.hip),
//   2. contracts  H = sum_k E_k' E_k,  E_k = sqrt(2 W_k) G_k[0:9],  straight from those registers
//      into register-resident accumulator tiles, and parks -H in LDS (tile = one b128 per lane),
//   3. runs a Mehrotra predictor-corrector interior-point method on
//         min 1/2 d'H d + g'd,  lo <= d <= hi      (d = U - Ubar, active thrusters only)
//      whose KKT matrix H + Sigma is factorised by a left-looking 16x16-blocked Cholesky that lives
//      ENTIRELY IN REGISTERS (tiles transposed, every tile product an MFMA on register operands,
//      every accumulator seeded directly from the LDS tile; diagonal tiles factorised and inverted
//      with v_readlane / fused DPP FMAs / v_permlane swaps), solved with packed VALU FMAs and
//      DPP / permlane reductions, and whose gradient follows the step through the solved Newton
//      system around one accurate float64 reference gradient (struct_grad),
//   4. leaves that iteration at mu 1e-5 for an ACTIVE-SET POLISH (the bounds with z > s as the set, two multiplier steps with a
//      diagonal penalty per round, signs verified; a round is a pass of the same loop) -- see the interior-point loop below.
// The algorithm is the one restated in oracle/qp_oracle.py (ipm_box) and oracle/ftmpc_oracle.c (ipm_box + polish_box); the reference solves
// the corresponding NLP with CasADi/IPOPT (ft_mpc/controllers/spiraling_mpc.py:87-238,319-354)
// followed by a cvxpy min-norm allocation (controllers/tools/control_allocator.py:65-94).
//
// LDS per wave (NB = tiles per dimension): NB(NB+1)/2 H tiles x 1 KiB + ~4 KiB
//   NB=8  (n<=128): 40.0 KiB -> 4 waves/CU (one per SIMD)
//   NB=9  (n<=144): 49.2 KiB -> 3 waves/CU
//   NB=10 (n<=160): 59.3 KiB -> 2 waves/CU
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- tile addressing -------------------------------------------------------------------
// Tile (I, J), I >= J, of the lower triangle is tile number I(I+1)/2 + J; in LDS a tile is 256 words
// in REGISTER ORDER (word 4*lane + s = register s of that lane): one conflict-free b128 per lane.
// One wave per workgroup: the LDS operations of a wave execute in issue order, so exchanging data
// between its lanes through LDS needs no s_barrier -- and, unlike __syncthreads(), must not drain the
// outstanding global loads (stage-record prefetch) or scratch traffic.  A compiler barrier suffices.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }
// same for the few exchanges that go through the per-workgroup global slot: stores must have landed
__device__ __forceinline__ void wave_global_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int tidx(int I, int J) { return (I * (I + 1)) / 2 + J; }

__device__ __forceinline__ float readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
// Whole-wave reductions, result in every lane: four DPP row rotations inside the 16-lane rows, then
// the two permlane swaps across the four rows (10 VALU, ~80 cycles; the __shfl_xor butterfly compiles to
// six dependent ds_bpermute round trips through the LDS crossbar, ~500 cycles per reduction).
struct OpAdd { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMin { static __device__ __forceinline__ float f(float a, float b) { return fminf(a, b); } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
template <class Op>
__device__ __forceinline__ float wave_reduce(float x) {
    x = Op::f(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false)));  // row_ror:8
    x = Op::f(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false)));  // row_ror:4
    x = Op::f(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false)));  // row_ror:2
    x = Op::f(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false)));  // row_ror:1
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // a = [r0,r1,r0,r1], b = [r2,r3,r2,r3]
    const float s = Op::f(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    unsigned c = __builtin_bit_cast(unsigned, s), d = c;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));   // c = [s0,s0,s2,s2], d = [s1,s1,s3,s3]
    return Op::f(__builtin_bit_cast(float, c), __builtin_bit_cast(float, d));
}
__device__ __forceinline__ float wave_sum(float x) { return wave_reduce<OpAdd>(x); }
__device__ __forceinline__ float wave_min(float x) { return wave_reduce<OpMin>(x); }
__device__ __forceinline__ float wave_max(float x) { return wave_reduce<OpMax>(x); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// The lane number, recomputed where it is needed (two v_mbcnt) and opaque to the optimiser.  Everything the hot regions
// derive from the lane -- li, lq, the per-lane LDS addresses of the vector workspace -- used to be computed once per
// kernel, kept live across the whole interior-point loop and SPILLED: every ds_write of a solution block then waited for a
// scratch reload of its own address (an HBM round trip in place of one v_lshlrev).
__device__ __forceinline__ int lane_now() {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}
// The Hessian tiles (register order: word 4*lane + s of a 256-word tile) live in LDS for NB = 8; for the
// larger instantiations they would cut the residency to 3 or 2 waves per CU, so there they sit in the
// per-workgroup global slot (L2 / Infinity-Cache resident; every lane re-reads only what it wrote itself).
typedef __attribute__((address_space(1))) float glb_f32;
// volatile read of an LDS byte table, typed with its address space (through a generic pointer the backend trips over the
// aperture test of the cast: "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
typedef __attribute__((address_space(3))) volatile unsigned char lds_vu8;
// NLDS = number of leading tiles (tile number I(I+1)/2 + J, i.e. whole block rows) kept in LDS; the
// rest sits in the global slot.  Tile numbers are compile-time constants at every call site.
// The global tiles are addressed through a BUFFER resource: the slot base sits in four SGPRs, the tile number is a scalar
// offset and every lane contributes the same 16 * lane bytes -- one VGPR for all tiles.  As plain pointers the compiler
// forms one 64-bit per-lane address per tile, hoists the lot out of the interior-point loop and spills it, so that every
// tile load waited for a scratch reload of its own address first.
template <int NLDS>
struct TileStore {
    float* p;                    // LDS, tiles [0, NLDS)
    __amdgpu_buffer_rsrc_t r;    // global slot, tiles [NLDS, ..)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    static constexpr bool is_global(int tile) { return tile >= NLDS; }
    static constexpr bool any_global(int ntiles) { return ntiles > NLDS; }
    __device__ __forceinline__ void bind(float* slot_tiles) {
        r = __builtin_amdgcn_make_buffer_rsrc(slot_tiles, 0, 0x7fffffff, 0x00020000);
    }
    __device__ __forceinline__ f32x4 ld(int tile, int lane) const {
        if (tile < NLDS) return *reinterpret_cast<const f32x4*>(p + tile * 256 + 4 * lane);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 16 * lane, (tile - NLDS) * 1024, 0));
    }
    __device__ __forceinline__ void st(int tile, int lane, f32x4 v) const {
        if (tile < NLDS) *reinterpret_cast<f32x4*>(p + tile * 256 + 4 * lane) = v;
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, 16 * lane, (tile - NLDS) * 1024, 0);
    }
};

// Diagnostic build only (-DFTMPC_STAMPS, never the shipped library): per-phase cycle totals
// of each instance's wave, written to SolveParams::dbg_H (reused as a u64 buffer).
#ifdef FTMPC_STAMPS
#define STAMP_DECL unsigned long long st_t0, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP_START() st_t0 = stamp_now()
#define STAMP(i)                                  \
    do {                                          \
        const unsigned long long st_t1 = stamp_now(); \
        st_acc[i] += st_t1 - st_t0;               \
        st_t0 = st_t1;                            \
    } while (0)
#else
#define STAMP_DECL
#define STAMP_START()
#define STAMP(i)
#endif

template <int NB>
struct Shape {
    static constexpr int NPAD = 16 * NB;
    static constexpr int NV = (NPAD + 63) / 64;
    static constexpr int NTILES = NB * (NB + 1) / 2;
    static constexpr int WORK = 2 * NPAD;                               // xvp | dvp
    static constexpr int SEXTRA = (NB == 9) ? 240 : 280;                                  // + stage storage of struct_grad (fills the 40 KiB/wave budget)
};

// ---- cross-lane primitives on the accumulator layout -------------------------------------
// Accumulator ("C") layout of a 16x16 tile: lane (q = lane>>4, col = lane&15) holds rows 4q..4q+3 of
// column col in 4 registers.  A 16-lane DPP row is one row-group q.
template <int N>
__device__ __forceinline__ float row_bcast(float x) {  // lane N of every 16-lane row -> whole row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + N, 0xf, 0xf, false));
}
// value of row-group QS (lanes 16QS..16QS+15) broadcast to all four row-groups, lane position kept:
// v_permlane16_swap duplicates within halves, v_permlane32_swap across halves (gfx950)
// The swap instructions exchange halves of TWO registers; both operands must be distinct
// registers even when they carry the same value, hence the opaque copy.
__device__ __forceinline__ unsigned opaque_copy(unsigned u) {
    unsigned v = u;
    asm volatile("" : "+v"(v));
    return v;
}
template <int QS>
__device__ __forceinline__ float group_bcast(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(u, opaque_copy(u), false, false);   // [r0,r0,r2,r2] , [r1,r1,r3,r3]
    const unsigned y = (QS & 1) ? r16[1] : r16[0];
    const auto r32 = __builtin_amdgcn_permlane32_swap(y, opaque_copy(y), false, false);   // [y0,y1,y0,y1] , [y2,y3,y2,y3]
    return __builtin_bit_cast(float, (QS & 2) ? r32[1] : r32[0]);
}
// Both results of one swap are needed for the quad sums; hipcc (ROCm 7.2) folds r[0] + r[1] of
// the builtin into r[0] + r[0], so the swap is issued through inline asm here (the s_nop covers
// the VALU-write -> permlane-read hazard that the compiler would otherwise pad).
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// sum over the four lanes that share lane&15 (one per row-group), result in all of them
__device__ __forceinline__ float quad_sum(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    swap32(a, b);                                                   // a = [r0,r1,r0,r1], b = [r2,r3,r2,r3]
    const float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    unsigned c = __builtin_bit_cast(unsigned, s), d = c;
    swap16(c, d);                                                   // c = [s0,s0,s2,s2], d = [s1,s1,s3,s3]
    return __builtin_bit_cast(float, c) + __builtin_bit_cast(float, d);
}
__device__ __forceinline__ double quad_sum_d(double x) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
    unsigned la = (unsigned)bits, lb = la, ha = (unsigned)(bits >> 32), hb = ha;
    swap32(la, lb);
    swap32(ha, hb);
    const double s = __builtin_bit_cast(double, ((unsigned long long)ha << 32) | la) +
                     __builtin_bit_cast(double, ((unsigned long long)hb << 32) | lb);
    const unsigned long long sb = __builtin_bit_cast(unsigned long long, s);
    unsigned lc = (unsigned)sb, ld = lc, hc = (unsigned)(sb >> 32), hd = hc;
    swap16(lc, ld);
    swap16(hc, hd);
    return __builtin_bit_cast(double, ((unsigned long long)hc << 32) | lc) +
           __builtin_bit_cast(double, ((unsigned long long)hd << 32) | ld);
}

// ---- 16x16 Cholesky + inverse of the diagonal tile, in the accumulator layout ---------------
// c[rr] = A[4q+rr][col] (full symmetric tile) on entry; on exit w[rr] = (L^-1)[4q+rr][col].
// Right-looking elimination WITHOUT scaling the pivot column: after step j column j of the tile is
// a_j = sqrt(d_j) L[:,j] and is never touched again by anything that matters, so the step is
//     c[r][col] += bcast_j(c[r]) * nm[col],   nm = -c[j][col] / d_j            (one v_fmac_f32_dpp per register)
// applied to EVERY lane: the lanes of the columns <= j are dead from here on (they are read only by
// the row_newbcast of their own step, which is over) and may hold anything, NaN included.
// The inverse is built alongside: E starts as I, row j of W is E[j]/sqrt(d_j), and
//     E[r][col] += bcast_j(c[r]) * nw[col],   nw = -E[j][col] / d_j            (c still unscaled -> 1/d, not 1/sqrt d)
// again on every lane (rows <= j of E are dead once row j has been copied out).  Per step: one
// v_readlane (pivot), one v_rsq, two row broadcasts through the permlane swaps, 8 fused DPP FMAs,
// one select.  No LDS, no barrier, no compare: a non-positive pivot d makes rsq(d) NaN or inf, every
// later nw = -E[j][col]/d has a 0 * inf in it, and W[15][15] (lane 63, register 3) ends up NaN (inf
// when only the last pivot is bad).
template <class F>
constexpr unsigned long long lane_mask(F f) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (f(l >> 4, l & 15)) m |= 1ull << l;
    return m;
}
// per lane: bit set ? if_set : if_clear.  The mask is a literal moved into vcc right here (two SALU
// issues next to a 4-cycle VALU are free); as "s" operands the ~100 distinct masks of the unrolled
// factorisation get hoisted out of the IPM loop and spilled, each use then pays two v_readlane.
template <unsigned long long M>
__device__ __forceinline__ float sel(float if_clear, float if_set) {
    float o;
    asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %4\n\tv_cndmask_b32_e32 %0, %1, %2, vcc"
        : "=v"(o)
        : "v"(if_clear), "v"(if_set), "n"((int)(unsigned)(M & 0xffffffffull)), "n"((int)(unsigned)(M >> 32))
        : "vcc");
    return o;
}
// d[r] += (lane J of the own 16-lane row of s[r]) * m   for the four registers of a tile.
// The leading s_nop covers "VALU writes VGPR -> DPP reads it" (2 wait states) for whatever the
// compiler scheduled right before; the four FMAs are independent of each other.
template <int J>
__device__ __forceinline__ void fmac4_rowbcast(float (&d)[4], const float (&s)[4], float m) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(m), "n"(J));
}
template <int J>
struct StepMasks {
    static constexpr unsigned long long grp = lane_mask([](int q, int) { return q == (J >> 2); });
    static constexpr unsigned long long piv = lane_mask([](int, int col) { return col == J; });
};
template <int J, class Work>
__device__ __forceinline__ void potrf_inv_step(float (&c)[4], float (&e)[4], float (&w)[4], int bperm_base, const Work& work) {
    constexpr int QJ = J >> 2, RJ = J & 3;
    // row J of E to every row-group through the LDS crossbar (ds_bpermute: no VALU slot, ~100 cycles of latency that
    // the pivot chain below covers); row J of the tile itself, which IS on the pivot chain, through the permlane swaps
    const float ej = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bperm_base + 64 * QJ, __builtin_bit_cast(int, e[RJ])));
    const float d = readlane_f(c[RJ], 16 * QJ + J);
    const float inv = __builtin_amdgcn_rsqf(d);
    const float nrd = -inv * inv;
    work.template run<5 * J + 0>();
    if constexpr (J < 15) {
        const float rowj = group_bcast<QJ>(c[RJ]);    // A[J][col] (= A[col][J]) in every row-group
        work.template run<5 * J + 1>();
        // the pivot column itself stays as it is (multiplier 0): the inverse below still needs it unscaled
        fmac4_rowbcast<J>(c, c, sel<StepMasks<J>::piv>(rowj * nrd, 0.f));
        work.template run<5 * J + 2>();
    } else {
        work.template run<5 * J + 1>();
        work.template run<5 * J + 2>();
    }
    // W[J][col] = E[J][col] / sqrt(d) into row-group QJ only: a DPP multiply with the identity lane pattern whose
    // row_mask enables just that row-group (multiply and select in one instruction; ej comes from the LDS
    // crossbar, not from a VALU write, so the DPP read hazard does not apply)
    // (the s_nop covers the wait state a VALU read needs after the transcendental v_rsq that produced `inv`, and the
    // two a DPP read needs should the compiler have copied `ej` with a VALU move: the hazard recognizer does not
    // look inside inline asm)
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:%3 bank_mask:0xf" : "+v"(w[RJ]) : "v"(ej), "v"(inv), "n"(1 << QJ));
    work.template run<5 * J + 3>();
    if constexpr (J < 15) fmac4_rowbcast<J>(e, c, ej * nrd);
    work.template run<5 * J + 4>();
}
struct NoWork {
    template <int S>
    __device__ __forceinline__ void run() const {}
};
template <class Work>
__device__ __forceinline__ void potrf_inv16(float (&c)[4], float (&w)[4], int lane, const Work& work) {
    const int q = lane >> 4, col = lane & 15;
    float e[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        e[rr] = (4 * q + rr == col) ? 1.f : 0.f;
        w[rr] = 0.f;
    }
    const int bb = 4 * col;   // ds_bpermute byte address of lane `col` of row-group 0
#ifndef FTMPC_POTRF_PRIO
#define FTMPC_POTRF_PRIO 1    // measured at two waves per SIMD: 13.96 -> 13.80 ms on the headline batch (0: off)
#endif
#if FTMPC_POTRF_PRIO
    __builtin_amdgcn_s_setprio(FTMPC_POTRF_PRIO);   // the pivot chain is dependency-bound: let it issue whenever it can, the partner wave's MFMA stream fills the rest
#endif
    potrf_inv_step<0>(c, e, w, bb, work);   potrf_inv_step<1>(c, e, w, bb, work);
    potrf_inv_step<2>(c, e, w, bb, work);   potrf_inv_step<3>(c, e, w, bb, work);
    potrf_inv_step<4>(c, e, w, bb, work);   potrf_inv_step<5>(c, e, w, bb, work);
    potrf_inv_step<6>(c, e, w, bb, work);   potrf_inv_step<7>(c, e, w, bb, work);
    potrf_inv_step<8>(c, e, w, bb, work);   potrf_inv_step<9>(c, e, w, bb, work);
    potrf_inv_step<10>(c, e, w, bb, work);  potrf_inv_step<11>(c, e, w, bb, work);
    potrf_inv_step<12>(c, e, w, bb, work);  potrf_inv_step<13>(c, e, w, bb, work);
    potrf_inv_step<14>(c, e, w, bb, work);  potrf_inv_step<15>(c, e, w, bb, work);
#if FTMPC_POTRF_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void potrf_inv16(float (&c)[4], float (&w)[4], int lane) {
    potrf_inv16(c, w, lane, NoWork{});
}
// The same for a diagonal tile whose rows 8..15 are identity padding (the last block of n = 16 (NB - 1) + 8 .. : the headline
// shape, n = 120): pivots 8..15 are 1 and their columns zero, so the elimination stops after eight steps and rows 8..15 of W
// are the untouched rows of E.  A bad pivot among the first eight shows as NaN / inf in W[7][7] (lane 23, register 3).
__device__ __forceinline__ void potrf_inv16_half(float (&c)[4], float (&w)[4], int lane) {
    const int q = lane >> 4, col = lane & 15;
    float e[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        e[rr] = (4 * q + rr == col) ? 1.f : 0.f;
        w[rr] = 0.f;
    }
    const int bb = 4 * col;
    const NoWork work{};
#if FTMPC_POTRF_PRIO
    __builtin_amdgcn_s_setprio(FTMPC_POTRF_PRIO);
#endif
    potrf_inv_step<0>(c, e, w, bb, work);   potrf_inv_step<1>(c, e, w, bb, work);
    potrf_inv_step<2>(c, e, w, bb, work);   potrf_inv_step<3>(c, e, w, bb, work);
    potrf_inv_step<4>(c, e, w, bb, work);   potrf_inv_step<5>(c, e, w, bb, work);
    potrf_inv_step<6>(c, e, w, bb, work);   potrf_inv_step<7>(c, e, w, bb, work);
#if FTMPC_POTRF_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) w[rr] = (q >= 2) ? e[rr] : w[rr];
}

// =============================================================================================
// Register-resident factorisation (all instantiations).  The factor never touches LDS: every tile
// lives in 4 registers per lane in the accumulator layout, and every tile product is
//     mm_tn(X, Y) = X' Y      (k = row index of both accumulator tiles: register s of
//                              row-group q' is k = 4q'+s, so both MFMA operands ARE registers)
// Kept per tile pair I > J:  T_IJ = L_IJ' ; per diagonal block W_J = L_JJ^-1 (Wd) and its transpose
// Wt_J (the T slot of the diagonal: operand of the panel solve T_IJ = W_J C_IJ').  The triangular
// solves run on the same registers with packed VALU FMAs (solve_reg).
// -H stays in the tile store untouched: every accumulator of the factorisation is seeded from it.
// =============================================================================================
__device__ __forceinline__ f32x4 mm_tn(const f32x4& X, const f32x4& Y, f32x4 acc) {
    acc = mfma4(X.x, Y.x, acc);
    acc = mfma4(X.y, Y.y, acc);
    acc = mfma4(X.z, Y.z, acc);
    acc = mfma4(X.w, Y.w, acc);
    return acc;
}

// A non-positive pivot leaves NaN or inf in W[15][15] (lane 63, .w), which the caller detects.
template <class Work>
__device__ __forceinline__ f32x4 potrf_inv16_call(f32x4 cin, int lane, const Work& work) {
    float c[4] = {cin.x, cin.y, cin.z, cin.w}, w[4];
    potrf_inv16(c, w, lane, work);
    const f32x4 r = {w[0], w[1], w[2], w[3]};
    return r;
}

// The off-diagonal Schur accumulations of block column J, b[I] += T_JK' T_IK (K < J < I): one MFMA per
// slot of potrf_inv16 (NM <= 48 for NB = 8), consecutive slots hit different accumulators.
template <int NB, int J>
struct SchurWork {
    static constexpr int NI = NB - 1 - J;
    static constexpr int NM = 4 * J * NI;            // MFMAs
    static_assert(NM <= 80, "more Schur MFMAs than potrf slots");
    const f32x4 (&T)[NB * (NB + 1) / 2];
    f32x4 (&b)[NB];
    template <int S>
    __device__ __forceinline__ void run() const {
        if constexpr (S < NM) {
            constexpr int K = S / (4 * NI), rem = S % (4 * NI), s4 = rem / NI, I = J + 1 + rem % NI;
            b[I] = mfma4(T[tidx(J, K)][s4], T[tidx(I, K)][s4], b[I]);
        }
    }
};

// Four row sums at once, written out: left to the compiler the DPP adds of neighbouring registers are
// paired into v_pk_add_f32, which cannot take a DPP operand, so every step becomes two v_mov_b32_dpp, a
// packed add and the moves that assemble the pairs (~3x the instructions).  Round-robin over the four
// registers: the 2 wait states a DPP read needs after a VALU write are filled by the other three adds.
__device__ __forceinline__ void row_sum16x4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row_sum16(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));  // row_ror:2
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));  // row_ror:1
    return x;
}

// one block column of the register-resident factorisation (template recursion instead of a
// `#pragma unroll` loop: the barrier inside would otherwise block the unroller and push the
// tile arrays into scratch).  T[tidx(I,J)] = L_IJ' for I > J, T[tidx(J,J)] = W_J', Wd[J] = W_J.
// WL: nullptr, or LDS for the inverse diagonal blocks W_J (NB x 256 words, register order) instead of Wd -- the two-waves-per-SIMD
// build of the NB = 8 instantiation keeps 28 tiles of the factor in registers and nothing else of it.
// PREF: tiles of the global slot are requested a block column ahead (registers for a whole column) or where they are used
// (a second resident wave covers the latency instead).
// LOUT: the tiles of the factor itself, L_JJ = C_JJ W_J' and L_IJ = C_IJ W_J', are written over the matrix tiles of the column
// just consumed (kernel 10 keeps L of the wrench-space Hessian in LDS this way).
// n_rows: the number of REAL rows of the matrix (n, not a block count and not a variable count of another space): rows at or
// beyond it are identity padding, and a last block with at most eight real rows takes the eight-pivot potrf.
template <int NB, int J, class TilesT, bool PREF = true, bool WLDS = false, bool LOUT = false>
__device__ __forceinline__ void chol_reg_col(const TilesT& tiles, const float* sigv, float* S, int n_rows, int lane, bool& ok,
                                             f32x4 (&T)[NB * (NB + 1) / 2], f32x4 (&Wd)[NB], const f32x4 (&pre)[NB], float* WL = nullptr) {
    lane = lane_now();
    const int li = lane & 15, lq = lane >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        // every accumulator starts from the stored tile (-H, register order) and collects  sum_K T_JK' T_IK  on top.
        // Tiles that live in the global slot were requested before the previous column's potrf (`pre`), and
        // the next column's are requested here, a potrf ahead of their use.
        f32x4 a0 = (PREF && TilesT::is_global(tidx(J, J))) ? pre[J] : tiles.ld(tidx(J, J), lane), a1 = zero;
        f32x4 bacc[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) bacc[I] = (I > J) ? ((PREF && TilesT::is_global(tidx(I, J))) ? pre[I] : tiles.ld(tidx(I, J), lane)) : zero;
        f32x4 nxt[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) nxt[I] = (PREF && J + 1 < NB && I > J && TilesT::is_global(tidx(I, J + 1))) ? tiles.ld(tidx(I, J + 1), lane) : zero;
#pragma unroll
        for (int K = 0; K < J; ++K) {
            if (K & 1) a1 = mm_tn(T[tidx(J, K)], T[tidx(J, K)], a1);
            else a0 = mm_tn(T[tidx(J, K)], T[tidx(J, K)], a0);
        }
        const float sg = sigv[16 * J + li];
        f32x4 cd;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) cd[rr] = ((4 * lq + rr == li) ? sg : 0.f) - (a0[rr] + a1[rr]);   // H + Sigma - sum
        f32x4 w;
        if (J == NB - 1 && n_rows <= 16 * (NB - 1) + 8) {      // rows 8 .. 15 of the last block are identity padding: eight pivots
            float ch[4] = {cd.x, cd.y, cd.z, cd.w}, wh[4];
            potrf_inv16_half(ch, wh, lane);
            w = f32x4{wh[0], wh[1], wh[2], wh[3]};
        } else {
            w = potrf_inv16_call(cd, lane, SchurWork<NB, J>{T, bacc});
        }
        ok = ok && (fabsf(w.w) <= 3.0e38f);   // NaN or inf in W[15][15] (W[7][7] for the half block): non-positive pivot
        if constexpr (WLDS) *reinterpret_cast<f32x4*>(WL + J * 256 + 4 * lane) = w;
        else Wd[J] = w;
        // Wt = W' through a 16x17 LDS scratch
        wave_lds_fence();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) S[(4 * lq + rr) * 17 + li] = w[rr];
        wave_lds_fence();
        f32x4 wt, wtn;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            wt[rr] = S[li * 17 + 4 * lq + rr];
            wtn[rr] = -wt[rr];
        }
        T[tidx(J, J)] = wt;
#pragma unroll
        for (int I = J + 1; I < NB; ++I) T[tidx(I, J)] = mm_tn(wtn, bacc[I], zero);    // L_IJ' = W_J (H_IJ' - sum) = -W_J bacc
        if constexpr (LOUT) {
            tiles.st(tidx(J, J), lane, mm_tn(cd, wt, zero));                             // C W' = L L' L^-T = L_JJ  (C symmetric)
#pragma unroll
            for (int I = J + 1; I < NB; ++I) tiles.st(tidx(I, J), lane, mm_tn(bacc[I], wtn, zero));   // (-C_IJ')' (-W_J') = C_IJ W_J' = L_IJ
        }
        if constexpr (J + 1 < NB) chol_reg_col<NB, J + 1, TilesT, PREF, WLDS, LOUT>(tiles, sigv, S, n_rows, lane, ok, T, Wd, nxt, WL);
    }
}

// the tiles of block column 0 that live in the global slot, requested by the caller ahead of the factorisation
template <int NB, class TilesT>
__device__ __forceinline__ void chol_prefetch_col0(const TilesT& tiles, int lane, f32x4 (&pre)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) pre[I] = TilesT::is_global(tidx(I, 0)) ? tiles.ld(tidx(I, 0), lane) : f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int NB, class TilesT, bool PREF = true, bool WLDS = false, bool LOUT = false>
__device__ __forceinline__ bool chol_reg(const TilesT& tiles, const float* sigv, float* S, int n_rows, int lane,
                                         f32x4 (&T)[NB * (NB + 1) / 2], f32x4 (&Wd)[NB], const f32x4 (&pre)[NB], float* WL = nullptr) {
    bool ok = true;
    chol_reg_col<NB, 0, TilesT, PREF, WLDS, LOUT>(tiles, sigv, S, n_rows, lane, ok, T, Wd, pre, WL);
    return __all(ok);
}

// xv: LDS vector in natural order; in: right-hand side, out: solution.  Matrix-vector work does not
// belong on the matrix cores (a vector as a 16-column tile wastes 15/16 of every MFMA and each
// dependent MFMA costs its full 8 passes): both sweeps are packed VALU FMAs on the register tiles,
// with the two reductions the accumulator layout offers,
//     over the 16 columns of a row-group (DPP row_ror)    -> "column tile" vectors  v[4q+s] in register s
//     over the 4 row-groups (permlane swaps)              -> "row" vectors          v[col] in every row-group
//   forward   r_J = b_J - sum_{K<J} L_JK y_K   (row vector:  sum_q sum_s T_JK[s] y_K[s]),   y_J = W_J r_J   (column tile)
//   backward  r_J = y_J - sum_{I>J} L_IJ' x_I  (column tile: row sums of T_IJ[s] x_I[col]), x_J = W_J' r_J  (row vector)
template <int NB, bool WLDS = false>
__device__ __forceinline__ void solve_reg(const f32x4 (&T)[NB * (NB + 1) / 2], const f32x4 (&WdR)[NB],
                                          float* xv, int nb, int lane, const float* WL = nullptr) {
    lane = lane_now();
    const int li = lane & 15, lq = lane >> 4;
    // the inverse diagonal blocks: registers, or (WL) one conflict-free b128 per block and sweep from LDS
    auto Wof = [&](int J) -> f32x4 {
        if constexpr (WLDS) return *reinterpret_cast<const f32x4*>(WL + J * 256 + 4 * lane);
        else return WdR[J];
    };
    // Right-looking order: as soon as a block of the solution is known its contribution goes to ALL
    // later right-hand sides.  These updates are independent of each other and fill the issue slots
    // of the one chain that is serial (reduce -> W -> reduce), which a single wave cannot hide otherwise.
    f32x4 Y[NB];
    f32x2 p[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) p[J] = f32x2{0.f, 0.f};
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        float r = xv[16 * J + li];
        if (J > 0) r -= quad_sum(p[J].x + p[J].y);
        {
            const f32x4 wj = Wof(J);
            float y0 = wj.x * r, y1 = wj.y * r, y2 = wj.z * r, y3 = wj.w * r;
            row_sum16x4(y0, y1, y2, y3);
            Y[J] = f32x4{y0, y1, y2, y3};
        }
#pragma unroll
        for (int I = J + 1; I < NB; ++I) {
            const f32x4& t = T[tidx(I, J)];
            p[I] += f32x2{t.x, t.y} * f32x2{Y[J].x, Y[J].y};
            p[I] += f32x2{t.z, t.w} * f32x2{Y[J].z, Y[J].w};
        }
    }
    wave_lds_fence();
    f32x2 a0[NB], a1[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) a0[J] = a1[J] = f32x2{0.f, 0.f};
    const int lb = lane_now();
    const int li_b = lb & 15, lq_b = lb >> 4;
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
        f32x4 r = Y[J];
        if (J < NB - 1) {
            float s0 = a0[J].x, s1 = a0[J].y, s2 = a1[J].x, s3 = a1[J].y;
            row_sum16x4(s0, s1, s2, s3);
            r -= f32x4{s0, s1, s2, s3};
        }
        const f32x4 wb = Wof(J);
        const f32x2 d2 = f32x2{wb.x, wb.y} * f32x2{r.x, r.y} + f32x2{wb.z, wb.w} * f32x2{r.z, r.w};
        const float xr = quad_sum(d2.x + d2.y);
        if (lq_b == 0) xv[16 * J + li_b] = xr;
        const f32x2 xx = {xr, xr};
#pragma unroll
        for (int K = 0; K < J; ++K) {
            const f32x4& t = T[tidx(J, K)];
            a0[K] += f32x2{t.x, t.y} * xx;
            a1[K] += f32x2{t.z, t.w} * xx;
        }
    }
    wave_lds_fence();
}

// =============================================================================================
// Accurate gradient at one reference point (float64, through the stage records instead of the
// fp32 Hessian):  grad(d) = 2 Bbar' Wbar (e_bar + Bbar d) + 2 Da' R (ut_bar + Da d) + 2 rho (u_bar + d)
// evaluated by a forward sweep  dc_{k+1} = A_k dc_k + B_k Da d_k  and an adjoint sweep
// lambda_k = A_k' lambda_{k+1} + W_k (e_k + dc_k),  g_k = 2 Da' (B_k' lambda_{k+1} + R(ut_k + Da d_k)) + 2 rho (..).
// Lanes 0..12 own one state row each; the stage record (152 doubles) sits in LDS next to the
// constants {0, 1, dt}.  After this call the IPM evaluates gradients as
//     grad(d) = grad(d_ref) + H32 (d - d_ref),
// so the fp32 rounding of H only acts on the (small) distance to the reference point.
// Cost: ~2 x N short dependent steps, once per instance (~2 % of a solve).
// =============================================================================================
__host__ __device__ constexpr int aoff(int r, int c) {   // record word of d(next state r)/d(input c); inputs: 13 state + 6 wrench
    constexpr int ZERO = REC_STRIDE, ONE = REC_STRIDE + 1, DT = REC_STRIDE + 2;
    if (r >= 13) return ZERO;
    const int kind = r < 3 ? 0 : (r < 6 ? 1 : (r < 9 ? 2 : 3));   // p, v, w, q row
    const int a = r < 3 ? r : (r < 6 ? r - 3 : (r < 9 ? r - 6 : r - 9));
    if (c < 3) return (kind == 0 && c == a) ? ONE : ZERO;                               // d/dp
    if (c < 6) return (kind == 0 && c - 3 == a) ? DT : ((kind == 1 && c - 3 == a) ? ONE : ZERO);   // d/dv
    if (c < 9) {                                                                         // d/dw
        const int j = c - 6;
        return kind == 0 ? REC_APW + 3 * a + j : (kind == 1 ? REC_AVW + 3 * a + j : (kind == 2 ? REC_AWW + 3 * a + j : REC_AQW + 3 * a + j));
    }
    if (c < 13) {                                                                        // d/dq
        const int j = c - 9;
        return kind == 0 ? REC_APQ + 4 * a + j : (kind == 1 ? REC_AVQ + 4 * a + j : (kind == 2 ? ZERO : REC_AQQ + 4 * a + j));
    }
    if (c < 16) {                                                                        // d/dF
        const int j = c - 13;
        return kind == 0 ? REC_BPF + 3 * a + j : (kind == 1 ? REC_BVF + 3 * a + j : ZERO);
    }
    const int j = c - 16;                                                                // d/dtau
    return kind == 0 ? REC_BPT + 3 * a + j : (kind == 1 ? REC_BVT + 3 * a + j : (kind == 2 ? REC_BWT + 3 * a + j : REC_BQT + 3 * a + j));
}
// the same as a table (16 rows x 19 inputs): evaluated per lane at run time the chain of selects above
// costs ~1000 instructions per call of struct_grad
struct AoffTable {
    unsigned char v[16][20];
};
constexpr AoffTable make_aoff_table() {
    AoffTable t{};
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 20; ++c) t.v[r][c] = (unsigned char)((c < 19) ? aoff(r, c) : REC_STRIDE);
    return t;
}
__device__ const AoffTable k_aoff = make_aoff_table();
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma_d(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// recg: this instance's float64 records; recd: LDS, REC_STRIDE+4 doubles; dnat: LDS, d in natural order
// (float); sS: (N+1) x 9 doubles of stage storage (LDS when it fits, else global); gout: global, n doubles.
// Result: gout[e] = 2 Bbar' Wbar (e_bar + Bbar d) + 2 Da' R (ut_bar + Da d)   (the caller adds 2 rho (u_bar + d)).
//
// Every matrix-vector product of the two sweeps is a v_mfma_f64_16x16x4_f64 chain: the 13-vector
// state/adjoint is kept as an accumulator "column tile" (lane (q, n) holds rows q, q+4, q+8, q+12,
// replicated over n), which is directly the B operand of the next product (k = 4s + q); the A
// operand M[m][4s+q] is read from the stage record in LDS through a per-lane word table.  No
// cross-lane traffic at all.
// The pointers carry their address space: as a non-inlined function with generic pointers every access
// became a FLAT instruction behind a run-time aperture test, and a flat access in flight forces every
// later wait to vmcnt(0) -- i.e. onto the record prefetch.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) const float lds_cf32;
typedef __attribute__((address_space(1))) double glb_f64;
typedef __attribute__((address_space(1))) const double glb_cf64;
// GOFF: offset (doubles) of the wrench scratch behind the gradient in the global slot (>= n)
template <class SPtr, int GOFF = 160>   // stage storage: lds_f64* when it fits behind the vectors, else glb_f64* (long horizons)
__device__ __noinline__ void struct_grad(const DeviceConsts& C, glb_cf64* recg, lds_f64* recd, lds_cf32* s_Da,
                                            lds_cf32* dnat, SPtr sS, glb_f64* gout, int na, int lane) {
    constexpr int ZERO = REC_STRIDE;
    const int N = C.N;
    const int m = lane & 15, q = lane >> 4;
    int offX[4], offG[2], offAT[4], offBT[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 4 * s4 + q;
        offX[s4] = (k < 13) ? k_aoff.v[m][k] : ZERO;                        // A[m][k]
        offAT[s4] = (k < 13 && m < 13) ? k_aoff.v[k][m] : ZERO;            // A[k][m]
        offBT[s4] = (k < 13 && m < 6) ? k_aoff.v[k][13 + m] : ZERO;        // B[k][m]
    }
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) offG[s4] = (4 * s4 + q < 6) ? k_aoff.v[m][13 + 4 * s4 + q] : ZERO;   // B[m][g]
    if (lane == 0) {
        recd[REC_STRIDE] = 0.0;
        recd[REC_STRIDE + 1] = 1.0;
        recd[REC_STRIDE + 2] = C.dt;
    }
    // per-lane rows of this lane's column tile: r_i = q + 4 i
    double qrow[4], rrow[2], da_op[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) qrow[i] = (q + 4 * i < 9) ? C.Q[q + 4 * i] : 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rrow[i] = (q + 4 * i < 6) ? C.R[q + 4 * i] : 0.0;
        da_op[i] = (q + 4 * i < 6) ? (double)s_Da[(q + 4 * i) * MAX_NT + m] : 0.0;   // Da[g][a=m]: A operand of Da' z
    }
    // terminal weight as an A operand, P[m][4s+q] (9x9, zero padded).  Read HERE: `C` is a generic
    // reference, and a flat load inside the sweeps would force every later wait to vmcnt(0).
    double pa[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) pa[s4] = (m < 9 && 4 * s4 + q < 9) ? C.P[9 * m + 4 * s4 + q] : 0.0;
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    // wrench perturbations gen_k = Da d_k of ALL stages up front, one (stage, component) pair per lane,
    // parked behind the gradient in the global scratch (N x 8 doubles) and prefetched with the records
    glb_f64* genS = gout + GOFF;
    for (int t = lane; t < N * 8; t += 64) {
        const int k = t >> 3, g = t & 7;
        double acc = 0.0;
        if (g < 6)
            for (int a = 0; a < na; ++a) acc += (double)s_Da[g * MAX_NT + a] * (double)dnat[k * na + a];
        genS[t] = acc;
    }
    __threadfence_block();
    wave_global_fence();   // lanes exchange through the global slot here
    // this lane's two wrench rows (g = q and q+4) of stage k
    const int g1 = (q + 4 < 6) ? q + 4 : 7;      // slot 7 of every stage is zero
    // The float64 records come from HBM (they were last touched by the condensing, a few hundred
    // thousand cycles ago) and one stage of a sweep is only ~500 cycles of work: the loads run FOUR
    // stages ahead, in four statically named register sets (a rotating copy would wait for the load).
    double pf[4][3], pg[4][2];
    // branch-free on purpose (clamped indices, a dump word for the lanes past the record): every
    // conditional around a load turns into a phi whose copy waits for the load just issued
    const int w2 = (lane + 128 < REC_STRIDE) ? lane + 128 : REC_STRIDE - 1;     // third word of this lane (152 = 2*64 + 24)
    const int d2 = (lane + 128 < REC_STRIDE) ? lane + 128 : REC_STRIDE + 3;     // LDS dump word behind {0, 1, dt}
    auto issue = [&](auto SL, int kk) {
        constexpr int sl = decltype(SL)::value;
        glb_cf64* r = recg + (int64_t)kk * REC_STRIDE;
        pf[sl][0] = r[lane];
        pf[sl][1] = r[lane + 64];
        pf[sl][2] = r[w2];
        pg[sl][0] = genS[kk * 8 + q];
        pg[sl][1] = genS[kk * 8 + g1];
    };
    auto stage_to_lds = [&](auto SL) {
        constexpr int sl = decltype(SL)::value;
        recd[lane] = pf[sl][0];
        recd[lane + 64] = pf[sl][1];
        recd[d2] = pf[sl][2];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    // ---- forward sweep: dc_{k+1} = A_k dc_k + B_k gen_k ----
    f64x4 dc = zero;
    auto fwd = [&](auto SL, int k) {
        constexpr int sl = decltype(SL)::value;
        wave_lds_fence();
        stage_to_lds(SL);
        const double g0 = pg[sl][0], g1v = pg[sl][1];
        issue(SL, (k + 4 < N) ? k + 4 : N - 1);
        wave_lds_fence();
        // two accumulators: a dependent float64 MFMA costs its full 16 passes, so halve the chain
        f64x4 nx = zero, ny = zero;
        nx = mfma_d(recd[offX[0]], dc.x, nx);
        ny = mfma_d(recd[offX[1]], dc.y, ny);
        nx = mfma_d(recd[offX[2]], dc.z, nx);
        ny = mfma_d(recd[offX[3]], dc.w, ny);
        nx = mfma_d(recd[offG[0]], g0, nx);
        ny = mfma_d(recd[offG[1]], g1v, ny);
        dc = nx + ny;
        // s_{k+1} = W (e_bar + dc)[0:9]: this lane's tile rows are q, q+4, q+8 (< 9 only for q = 0), q+12
        f64x4 t = zero;
        if (k + 1 == N) {   // terminal weight P: A operand P[m][4s+q] (9x9, zero padded)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) t = mfma_d(pa[s4], dc[s4], t);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) t[i] = qrow[i] * dc[i];
        }
        if (m == 0) {
            SPtr srow = sS + (k + 1) * 9;
            srow[q] = recd[REC_WE + q] + t.x;
            srow[q + 4] = recd[REC_WE + q + 4] + t.y;
            if (q == 0) srow[8] = recd[REC_WE + 8] + t.z;
        }
    };
    const int N4 = N & ~3;
    issue(S0{}, 0);
    issue(S1{}, (1 < N) ? 1 : 0);
    issue(S2{}, (2 < N) ? 2 : 0);
    issue(S3{}, (3 < N) ? 3 : 0);
    for (int k = 0; k < N4; k += 4) {
        fwd(S0{}, k);
        fwd(S1{}, k + 1);
        fwd(S2{}, k + 2);
        fwd(S3{}, k + 3);
    }
    for (int k = N4; k < N; ++k) {   // N not a multiple of 4: the tail reloads its stage (latency exposed)
        issue(S0{}, k);
        fwd(S0{}, k);
    }
    wave_global_fence();   // the stage storage may be the global slot (long horizons)
    // ---- adjoint sweep (stage N-1-i uses register set i & 3) ----
    f64x4 lam = zero;
    auto adj = [&](auto SL, int k) {
        constexpr int sl = decltype(SL)::value;
        wave_lds_fence();
        stage_to_lds(SL);
        const double g0 = pg[sl][0], g1v = pg[sl][1];
        issue(SL, (k - 4 >= 0) ? k - 4 : 0);
        wave_lds_fence();
        if (k == N - 1) {   // lambda_N = s_N
            lam.x = sS[N * 9 + q];
            lam.y = sS[N * 9 + q + 4];
            lam.z = (q == 0) ? sS[N * 9 + 8] : 0.0;
            lam.w = 0.0;
        }
        // z = B_k' lambda_{k+1} + R (ut_bar + Da d_k): rows g = q, q+4 (< 6)
        f64x4 z = zero, zb = zero;
        z = mfma_d(recd[offBT[0]], lam.x, z);
        zb = mfma_d(recd[offBT[1]], lam.y, zb);
        z = mfma_d(recd[offBT[2]], lam.z, z);
        zb = mfma_d(recd[offBT[3]], lam.w, zb);
        z += zb;
        const double z0 = z.x + recd[REC_RUT + q] + rrow[0] * g0;                              // row q  (< 4 <= 6)
        const double z1 = (q + 4 < 6) ? z.y + recd[REC_RUT + q + 4] + rrow[1] * g1v : 0.0;     // row q+4
        // g_{k,a} = 2 Da[:,a]' z : A operand Da[g = 4s+q][a = m], B operand z rows 4s+q
        f64x4 ga = zero;
        ga = mfma_d(da_op[0], z0, ga);
        ga = mfma_d(da_op[1], z1, ga);
        if (m == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (q + 4 * i < na) gout[k * na + q + 4 * i] = 2.0 * ga[i];
        }
        // lambda_k = A_k' lambda_{k+1} + [s_k; 0]
        if (k > 0) {
            f64x4 nl = zero, nm = zero;
            nl = mfma_d(recd[offAT[0]], lam.x, nl);
            nm = mfma_d(recd[offAT[1]], lam.y, nm);
            nl = mfma_d(recd[offAT[2]], lam.z, nl);
            nm = mfma_d(recd[offAT[3]], lam.w, nm);
            nl += nm;
            lam.x = nl.x + sS[k * 9 + q];
            lam.y = nl.y + sS[k * 9 + q + 4];
            lam.z = nl.z + ((q == 0) ? sS[k * 9 + 8] : 0.0);
            lam.w = nl.w;
        }
    };
    issue(S0{}, N - 1);
    issue(S1{}, (N - 2 >= 0) ? N - 2 : 0);
    issue(S2{}, (N - 3 >= 0) ? N - 3 : 0);
    issue(S3{}, (N - 4 >= 0) ? N - 4 : 0);
    int ka = N - 1;
    for (; ka >= 3; ka -= 4) {
        adj(S0{}, ka);
        adj(S1{}, ka - 1);
        adj(S2{}, ka - 2);
        adj(S3{}, ka - 3);
    }
    for (; ka >= 0; --ka) {
        issue(S0{}, ka);
        adj(S0{}, ka);
    }
    __threadfence_block();
    wave_global_fence();   // lanes exchange through the global slot here
}

// ---- dense operand images of the stage matrices ------------------------------------------------
// State component r (p 0-2, v 3-5, w 6-8, q 9-12) -> row of the 16-row sensitivity tile.  The nine
// costed components go to rows {4q+s : q, s < 3} (registers 0..2 of row-groups 0..2), so that
// E' E contracts over three MFMAs; the quaternion takes rows 12, 13, 14 and 3.
__host__ __device__ constexpr int tile_row(int r) { return r < 9 ? 4 * (r % 3) + r / 3 : (r < 12 ? 12 + (r - 9) : 3); }
constexpr int DENSE_DUMP = 256 + 128;          // write-only word for record entries that are not matrix entries
constexpr int DENSE_WORDS = 400;               // A (256) | B (128) | dump, padded
constexpr int dense_blk(int w, int base, int ncol, int rowc, int colc, bool isB) {
    const int a = (w - base) / ncol, c = (w - base) % ncol;
    return isB ? 256 + 8 * tile_row(rowc + a) + (colc + c) : 16 * tile_row(rowc + a) + tile_row(colc + c);
}
constexpr int dense_pos(int w) {
    if (w < REC_APQ) return dense_blk(w, REC_APW, 3, 0, 6, false);
    if (w < REC_AVW) return dense_blk(w, REC_APQ, 4, 0, 9, false);
    if (w < REC_AVQ) return dense_blk(w, REC_AVW, 3, 3, 6, false);
    if (w < REC_AWW) return dense_blk(w, REC_AVQ, 4, 3, 9, false);
    if (w < REC_AQW) return dense_blk(w, REC_AWW, 3, 6, 6, false);
    if (w < REC_AQQ) return dense_blk(w, REC_AQW, 3, 9, 6, false);
    if (w < REC_BPF) return dense_blk(w, REC_AQQ, 4, 9, 9, false);
    if (w < REC_BPT) return dense_blk(w, REC_BPF, 3, 0, 0, true);
    if (w < REC_BVF) return dense_blk(w, REC_BPT, 3, 0, 3, true);
    if (w < REC_BVT) return dense_blk(w, REC_BVF, 3, 3, 0, true);
    if (w < REC_BWT) return dense_blk(w, REC_BVT, 3, 3, 3, true);
    if (w < REC_BQT) return dense_blk(w, REC_BWT, 3, 6, 3, true);
    if (w < REC_WE) return dense_blk(w, REC_BQT, 3, 9, 3, true);
    return DENSE_DUMP;
}
struct DensePosTable {
    unsigned short v[REC_STRIDE];
};
constexpr DensePosTable make_dense_pos() {
    DensePosTable t{};
    for (int w = 0; w < REC_STRIDE; ++w) t.v[w] = (unsigned short)dense_pos(w);
    return t;
}
__device__ const DensePosTable k_dense_pos = make_dense_pos();
// compile-time checks of the scatter: every Jacobian word has its own cell, none lands on a constant
// entry (I, dt I) of the A image, and the non-matrix words (W e, R ut, padding) go to the dump word
constexpr bool dense_pos_ok() {
    bool used[DENSE_WORDS] = {};
    for (int a = 0; a < 3; ++a) {
        used[16 * tile_row(a) + tile_row(a)] = true;          // d p / d p
        used[16 * tile_row(a) + tile_row(3 + a)] = true;      // d p / d v = dt
        used[16 * tile_row(3 + a) + tile_row(3 + a)] = true;  // d v / d v
    }
    for (int w = 0; w < REC_STRIDE; ++w) {
        const int p = dense_pos(w);
        if (w >= REC_WE) {
            if (p != DENSE_DUMP) return false;
            continue;
        }
        if (p < 0 || p >= DENSE_DUMP || used[p]) return false;
        used[p] = true;
    }
    return true;
}
static_assert(dense_pos_ok(), "stage-record scatter table is inconsistent");
constexpr bool tile_row_ok() {
    bool seen[16] = {};
    for (int r = 0; r < 13; ++r) {
        const int t = tile_row(r);
        if (t < 0 || t > 15 || seen[t]) return false;
        seen[t] = true;
        if (r < 9 && ((t & 3) > 2 || (t >> 2) > 2)) return false;   // costed components: registers 0..2 of row-groups 0..2
    }
    return true;
}
static_assert(tile_row_ok(), "state-component permutation of the sensitivity tiles is inconsistent");

}  // namespace

// Resident waves per SIMD of the NB = 8 instantiation (the headline shape: n <= 128).  2: 256 registers and <= 20 KiB of LDS per
// wave -- Hessian tiles in the per-workgroup global slot, W_J in LDS, see OCC2 below; 1: the round-2 form (512 registers, all
// tiles in LDS).  Measured, 65 536 double-fault instances: 17.5 ms at 1, 14.1-15.0 ms at 2.
#ifndef FTMPC_F32_MU_POLISH
#define FTMPC_F32_MU_POLISH 1e-5f      // leave the interior-point iteration for the active-set polish below this mu (0: never)
#endif
#ifndef FTMPC_F32_REFINE_AT_POLISH
#define FTMPC_F32_REFINE_AT_POLISH 1   // the float64 gradient is taken inside the polish, between its two steps (not at mu_refine before it)
#endif
#ifndef FTMPC_F32_REGRAD_DD
#define FTMPC_F32_REGRAD_DD 2e-4f      // polish rounds after the first: another float64 gradient when the round's first step exceeds this (x ub)
#endif
#ifndef FTMPC_F32_PW0
#define FTMPC_F32_PW0 1e3f             // penalty of the polish over max diag(H)
#endif
#ifndef FTMPC_F32_OCC
#define FTMPC_F32_OCC 2
#endif
template <int NB>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((NB == 8) ? FTMPC_F32_OCC : 1, (NB == 8) ? FTMPC_F32_OCC : 1))) ftmpc_solve_f32_kernel(const DeviceConsts C, const SolveParams P) {
    using SH = Shape<NB>;
    constexpr int NPAD = SH::NPAD, NV = SH::NV, NTILES = SH::NTILES;
    // Hessian tiles kept in LDS (whole block rows; see TileStore): all 36 for NB = 8; the first 8 block rows
    // (36 tiles) for NB = 9, whose ninth row comes from the global slot one tile per column; the first 35
    // of 55 for NB = 10.  Each choice leaves 4 resident waves per CU (one per SIMD).
    constexpr int NLDS = (NB == 8) ? ((FTMPC_F32_OCC > 1) ? 0 : NTILES) : (NB == 9 ? 36 : 35);
    // NB = 8 built for TWO resident waves per SIMD (FTMPC_F32_OCC = 2): 256 registers and 20 KiB of LDS per wave.  The Hessian
    // tiles go to the per-workgroup global slot (Infinity-Cache resident, read where they are used: the partner wave covers the
    // latency), the inverse diagonal blocks W_J to LDS, and the build keeps no scaled copy of the sensitivity tiles.
    constexpr bool OCC2 = (NB == 8) && (FTMPC_F32_OCC > 1);
    constexpr int BUILD_WORDS = 2 * DENSE_WORDS + 256;   // dense stage-matrix images of the build phase; later N*NT output words
    __shared__ __attribute__((aligned(16))) float tiles[(NLDS * 256 > BUILD_WORDS) ? NLDS * 256 : BUILD_WORDS];
    __shared__ __attribute__((aligned(16))) float recbuf[2 * REC_STRIDE + 8];   // two fp32 stage records | one fp64 record + {0,1,dt}
    __shared__ __attribute__((aligned(16))) float work[SH::WORK + SH::SEXTRA];
    __shared__ __attribute__((aligned(16))) float s_Da[6 * MAX_NT];
    __shared__ unsigned char s_stg[NPAD], s_thr[NPAD];
    __shared__ int s_act[MAX_NT];
    __shared__ __attribute__((aligned(16))) float wlds[OCC2 ? NB * 256 : 4];   // W_J of the current factorisation (OCC2)
    float* const dense = tiles;      // build phase only: 2 x (A 16x16 | B 16x8 | dump word), then LPt 16x16
    float* const xvp = work;         // rhs / solution of the KKT solves
    float* const dvp = work + NPAD;  // d, permuted layout (gradient mat-vec); with the tail: struct_grad stage storage

    const int lane = threadIdx.x;
    const int li = lane & 15, lq = lane >> 4;
    const int N = C.N, NT = C.NT;
    TileStore<NLDS> htiles;
    htiles.p = tiles;
    htiles.bind(P.hscratch + (int64_t)blockIdx.x * P.tile_words + slot_tile_off_words(C.N));
    const float rho = (float)C.rho;
    const float mu_stop = (float)C.mu_stop;
    float Rf[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) Rf[g] = (float)C.R[g];

    // the waves pull their instances from this launch's work list (ftmpc_linearize.hip) through a shared cursor; the
    // next number is requested while the current instance is being solved
    auto pull = [&]() {
        int i = 0;
        if (lane == 0) i = atomicAdd(P.qhead, 1);
        return i;
    };
    const int qn = *P.qcount;
    int qnext = pull();
    for (;;) {
        const int qi = __builtin_amdgcn_readfirstlane(qnext);
        if (qi >= qn) break;
        const int64_t inst = P.qlist[qi];
        qnext = pull();
        STAMP_DECL;
        STAMP_START();
        wave_lds_fence();
        // ---------------- prologue: active thrusters, index tables ----------------
        double ub_l = 0.0;
        if (lane < NT) ub_l = P.ub[inst * NT + lane];
        const unsigned long long amask = __ballot(lane < NT && ub_l > 0.0);
        const int na = __popcll(amask);
        const int n = N * na;
        const int nb = (n + 15) >> 4;
        const int npad = nb * 16;
        (void)npad;
        if (na == 0 || nb > NB) {  // nothing to optimise / shape not supported by any instantiation
            if (lane < NT) P.out_u0[inst * NT + lane] = 0.0;
            if (P.out_U)
                for (int i = lane; i < N * NT; i += 64) P.out_U[inst * N * NT + i] = 0.0;
            if (lane == 0) {
                if (P.status) P.status[inst] = (na == 0) ? 0 : 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        // the register-resident path always runs the full NB x NB tile grid (identity padding blocks),
        // which keeps its factorisation free of data-dependent branches
        const int nbr = NB;
        const int npadr = 16 * nbr;
        const int myrank = __popcll(amask & ((1ull << lane) - 1ull));
        if (lane < NT && ub_l > 0.0) s_act[myrank] = lane;
        wave_lds_fence();
        if (lane < na) {
#pragma unroll
            for (int g = 0; g < 6; ++g) s_Da[g * MAX_NT + lane] = (float)C.D[g * MAX_NT + s_act[lane]];
        }
        for (int e = lane; e < npadr; e += 64) {
            const int s = e / na;
            s_stg[e] = (unsigned char)(e < n ? s : 255);
            s_thr[e] = (unsigned char)(e < n ? e - s * na : 255);
        }
        // stage records are float64 (ftmpc_linearize.hip); the condensing runs on their fp32 rounding,
        // the reference-point gradient (struct_grad) on the full values
        typedef double f64x4_t __attribute__((ext_vector_type(4)));
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        // prefetched one stage ahead and kept RAW: converting right after the load would put the whole
        // global latency on the stage's critical path
        f64x4_t pre64 = {0.0, 0.0, 0.0, 0.0};
        if (lane < REC_STRIDE / 4) pre64 = *reinterpret_cast<const f64x4_t*>(recg + 4 * lane);
        // where this lane's four record words go in the dense images (loaded per instance: as a kernel-long
        // live range the four values end up in scratch and every stage waits for them)
        int dpos[4] = {DENSE_DUMP, DENSE_DUMP, DENSE_DUMP, DENSE_DUMP};
        {
            int w0 = 4 * lane;
            asm volatile("" : "+v"(w0));
            if (lane < REC_STRIDE / 4) {
                const ushort4 t = *reinterpret_cast<const ushort4*>(&k_dense_pos.v[w0]);
                dpos[0] = t.x;
                dpos[1] = t.y;
                dpos[2] = t.z;
                dpos[3] = t.w;
            }
        }
        // dense operand images of the stage matrices (build phase only, in the tile area):
        //   dense[b] = A_k as 16x16 (rows/columns permuted by tile_row) | [B_pF B_pT; ...] as 16x8
        // constant entries (I, dt I) and zeros are written here once, the record scatter keeps them.
        for (int i = lane; i < 2 * DENSE_WORDS + 256; i += 64) dense[i] = 0.f;
        wave_lds_fence();
        if (lane < 6) {
            const int a = lane % 3, isv = lane / 3;              // rows p_a (isv = 0) and v_a (isv = 1)
#pragma unroll
            for (int bsel = 0; bsel < 2; ++bsel) {
                float* dd = dense + bsel * DENSE_WORDS;
                dd[16 * tile_row(3 * isv + a) + tile_row(3 * isv + a)] = 1.f;
                if (!isv) dd[16 * tile_row(a) + tile_row(3 + a)] = (float)C.dt;
            }
        }
        for (int i = lane; i < 81; i += 64) {
            const int r = i / 9, c = i - 9 * r;
            if (c >= r) dense[2 * DENSE_WORDS + 16 * tile_row(r) + tile_row(c)] = (float)C.LPt[i];
        }
        wave_lds_fence();
        const f32x4 lp4 = lds4(dense + 2 * DENSE_WORDS + 16 * li + 4 * lq);   // A operand of E_N = LPt G9

        // per-lane column bookkeeping: column e = v*64 + lane
        int kcol[NV], acol[NV];
        float ubar[NV], ubv[NV], gacc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = v * 64 + lane;
            kcol[v] = (e < npadr) ? s_stg[e] : 255;
            acol[v] = (e < npadr) ? s_thr[e] : 255;
            ubar[v] = 0.f;
            ubv[v] = 1.f;
            gacc[v] = 0.f;
            if (kcol[v] != 255) {
                const int t = s_act[acol[v]];
                ubv[v] = (float)P.ub[inst * NT + t];
                if (P.warmU) {
                    const float wv = (float)P.warmU[(inst * N + kcol[v]) * NT + t];
                    ubar[v] = fminf(fmaxf(wv, 0.f), ubv[v]);
                }
            }
        }
        // the sensitivity G = d c_{k+1} / d U as accumulator tiles: G[X] reg s of lane (q, col) is row 4q+s
        // (tile_row order: the 9 costed components sit in registers 0..2 of row-groups 0..2) of column 16X+col
        int kX[OCC2 ? 1 : NB];
        float DaB[OCC2 ? 1 : NB][2];           // B operand of the new columns: Da[g = 4s+q][thruster of column 16X+col]
        f32x4 G[NB];
        float gpart[NB];
#pragma unroll
        for (int X = 0; X < NB; ++X) {
            if constexpr (!OCC2) {
                kX[X] = s_stg[16 * X + li];
                const int ax = s_thr[16 * X + li];
                DaB[X][0] = (ax != 255) ? s_Da[lq * MAX_NT + ax] : 0.f;
                DaB[X][1] = (ax != 255 && lq < 2) ? s_Da[(4 + lq) * MAX_NT + ax] : 0.f;
            }
            G[X] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gpart[X] = 0.f;
        }
        float esc[3];               // sqrt(2 Q) of this lane's costed rows (row-group 3 holds the quaternion: 0); OCC2: 2 Q
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) esc[s3] = (lq < 3) ? (OCC2 ? 2.f * (float)C.Q[3 * s3 + lq] : (float)C.sq2Q[3 * s3 + lq]) : 0.f;
        f32x4 acc[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // finish tile (I,J): + 2 (Da' R Da + rho I) on the blocks of equal stage, unit pad diagonal, then -H to the tile store,
        // each tile exactly as its lanes hold it (one b128 per lane, conflict free): the factorisation loads a tile STRAIGHT
        // INTO the MFMA accumulator that collects the Schur terms (sum T'T - H), so no VALU instruction touches it.
        float* const mtab = work;     // MAX_NT x MAX_NT words; the vector workspace is idle until the interior-point iterations
        auto finish_tile = [&](int I, int J) {     // I, J are constants after unrolling
            f32x4 h = acc[(I * (I + 1)) / 2 + J];
            if (J >= I - 1) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    // register rr of lane (lq, li): H[16I + 4lq+rr][16I + li] on the diagonal, H[16I + li][16J + 4lq+rr] below it
                    const int e1 = (I == J) ? 16 * I + 4 * lq + rr : 16 * I + li;
                    const int e2 = (I == J) ? 16 * J + li : 16 * J + 4 * lq + rr;
                    const int s1 = s_stg[e1], a1 = s_thr[e1];
                    const int s2 = s_stg[e2], a2 = s_thr[e2];
                    // M[a1][a2] = 2 (sum_g Da[g][a1] R_g Da[g][a2] + rho [a1 == a2]), tabulated below (one read instead of twelve)
                    float add = (s1 != 255 && s1 == s2) ? mtab[((a1 & (MAX_NT - 1)) << 4) | (a2 & (MAX_NT - 1))] : 0.f;
                    if (s1 == 255 && e1 == e2) add = 1.f;
                    h[rr] += add;
                }
            }
            htiles.st((I * (I + 1)) / 2 + J, lane, -h);
        };

        STAMP(0);
        // ---------------- build: stage loop (everything on the matrix cores) ----------------
        //   G_X <- A_k G_X (+ B_k Da on the columns of stage k),  E_X = sqrt(2W) G_X (rows 0..8),
        //   acc(I,J) += E_I' E_J,   g += G' W e_k
        auto stage = [&](int k, auto TERM) {
            constexpr bool terminal = decltype(TERM)::value;
            const int lane = lane_now();
            const int li = lane & 15, lq = lane >> 4;
            float* rb = recbuf + (k & 1) * REC_STRIDE;
            float* dd = dense + (k & 1) * DENSE_WORDS;
            if (lane < REC_STRIDE / 4) {
                const f32x4 pre = {(float)pre64.x, (float)pre64.y, (float)pre64.z, (float)pre64.w};
                *reinterpret_cast<f32x4*>(rb + 4 * lane) = pre;
                dd[dpos[0]] = pre.x;
                dd[dpos[1]] = pre.y;
                dd[dpos[2]] = pre.z;
                dd[dpos[3]] = pre.w;
                if (!terminal) pre64 = *reinterpret_cast<const f64x4_t*>(recg + (k + 1) * REC_STRIDE + 4 * lane);
            }
            wave_lds_fence();
            const f32x4 a4 = lds4(dd + 16 * li + 4 * lq);
            const float b0 = dd[256 + 8 * li + lq], b1 = dd[256 + 8 * li + 4 + lq];
            float we[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) we[s3] = (lq < 3) ? rb[REC_WE + 3 * s3 + lq] : 0.f;
            const float rut0 = rb[REC_RUT + lq], rut1 = (lq < 2) ? rb[REC_RUT + 4 + lq] : 0.f;
            const int Imax = ((k + 1) * na - 1) >> 4;
            const int Xnew = (k * na) >> 4;
            float E[NB][3];
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int X = 0; X < NB; ++X)
                if (X <= Imax) {
                    f32x4 o = zero4;
                    if (X >= Xnew) {
                        float d0, d1;
                        if constexpr (OCC2) {     // the one or two tiles that hold the columns of stage k: read off the tables
                            const int ax = *(lds_vu8*)&s_thr[16 * X + li];
                            const bool mine = (*(lds_vu8*)&s_stg[16 * X + li] == k);
                            d0 = mine ? s_Da[lq * MAX_NT + (ax & (MAX_NT - 1))] : 0.f;
                            d1 = (mine && lq < 2) ? s_Da[(4 + lq) * MAX_NT + (ax & (MAX_NT - 1))] : 0.f;
                        } else {
                            const bool mine = (kX[X] == k);
                            d0 = mine ? DaB[X][0] : 0.f;
                            d1 = mine ? DaB[X][1] : 0.f;
                        }
                        o = mfma4(b0, d0, o);
                        o = mfma4(b1, d1, o);
                        gpart[X] += d0 * rut0 + d1 * rut1;   // input-cost gradient of the new columns, Da[:, a]' (R .* ut_k): rows g = lq, 4+lq
                    }
                    o = mfma4(a4.x, G[X].x, o);
                    o = mfma4(a4.y, G[X].y, o);
                    o = mfma4(a4.z, G[X].z, o);
                    o = mfma4(a4.w, G[X].w, o);
                    G[X] = o;
                }
#pragma unroll
            for (int X = 0; X < NB; ++X)
                if (X <= Imax) {
                    gpart[X] += we[0] * G[X].x + we[1] * G[X].y + we[2] * G[X].z;
                    if (!terminal) {
                        if constexpr (!OCC2) {
                            E[X][0] = esc[0] * G[X].x;
                            E[X][1] = esc[1] * G[X].y;
                            E[X][2] = esc[2] * G[X].z;
                        }
                    } else {
                        f32x4 o = zero4;
                        o = mfma4(lp4.x, G[X].x, o);
                        o = mfma4(lp4.y, G[X].y, o);
                        o = mfma4(lp4.z, G[X].z, o);
                        o = mfma4(lp4.w, G[X].w, o);
                        E[X][0] = o.x;
                        E[X][1] = o.y;
                        E[X][2] = o.z;
                    }
                }
            STAMP(1);
#pragma unroll
            for (int I = 0; I < NB; ++I)
                if (I <= Imax) {
                    float eI[3] = {0.f, 0.f, 0.f};
                    if constexpr (OCC2 && !terminal) {   // E_J' E_I = G_J' (2 Q) G_I: the weight on one operand, no scaled copy of G
                        eI[0] = esc[0] * G[I].x;
                        eI[1] = esc[1] * G[I].y;
                        eI[2] = esc[2] * G[I].z;
                    }
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        // tile (I,J) is kept TRANSPOSED, (H_IJ)' = E_J' E_I: the layout the factorisation consumes
                        if constexpr (OCC2 && !terminal) {
                            acc[(I * (I + 1)) / 2 + J] = mfma4(G[J].x, eI[0], acc[(I * (I + 1)) / 2 + J]);
                            acc[(I * (I + 1)) / 2 + J] = mfma4(G[J].y, eI[1], acc[(I * (I + 1)) / 2 + J]);
                            acc[(I * (I + 1)) / 2 + J] = mfma4(G[J].z, eI[2], acc[(I * (I + 1)) / 2 + J]);
                        } else {
#pragma unroll
                        for (int s3 = 0; s3 < 3; ++s3) acc[(I * (I + 1)) / 2 + J] = mfma4(E[J][s3], E[I][s3], acc[(I * (I + 1)) / 2 + J]);
                        }
                        // last stage (peeled copy of the stage code): the tile is complete -- finish and store it right
                        // here, so that its accumulator dies now; finishing all 36 after the loop keeps them live across
                        // that code and costs ~50 scratch round trips
                        if constexpr (terminal) finish_tile(I, J);
                    }
                }
            STAMP(2);
        };
        for (int k = 0; k + 1 < N; ++k) stage(k, std::false_type{});
        for (int t = lane; t < MAX_NT * MAX_NT; t += 64) {
            const int a1 = t >> 4, a2 = t & (MAX_NT - 1);
            float m = 0.f;
            if (a1 < na && a2 < na) {
#pragma unroll
                for (int g = 0; g < 6; ++g) m += s_Da[g * MAX_NT + a1] * Rf[g] * s_Da[g * MAX_NT + a2];
                if (a1 == a2) m += rho;
            }
            mtab[t] = 2.f * m;
        }
        wave_lds_fence();
        stage(N - 1, std::true_type{});
        // block rows beyond the last stage's columns (tiny problems only): pure identity padding
#pragma unroll
        for (int I = 0; I < NB; ++I)
            if (I > ((N * na - 1) >> 4)) {
#pragma unroll
                for (int J = 0; J <= I; ++J) finish_tile(I, J);
            }
        // column gradients back to the column-per-lane order: column 64v + lane sits in tile 4v + lq
        {
            float qs[NB];
#pragma unroll
            for (int X = 0; X < NB; ++X) qs[X] = quad_sum(gpart[X]);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * v + j < NB) t = (lq == j) ? qs[4 * v + j] : t;
                gacc[v] += t;
            }
        }
        wave_lds_fence();   // the dense images in the tile area are dead from here

        // g, bounds, start point
        float gv[NV], lo[NV], hi[NV], sl[NV], su[NV], zl[NV], zu[NV], grad[NV];
        bool valid[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            valid[v] = kcol[v] != 255;
            gv[v] = valid[v] ? 2.f * (gacc[v] + rho * ubar[v]) : 0.f;
            lo[v] = -ubar[v];
            hi[v] = ubv[v] - ubar[v];
            sl[v] = su[v] = 0.5f * ubv[v];
        }
        if (P.dbg_inst == inst) {  // test hook: dump the QP this wave is about to solve
#pragma unroll
            for (int I = 0; I < NB; ++I)
                if (I < nbr) {
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        const f32x4 ht = htiles.ld((I * (I + 1)) / 2 + J, lane);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int e1 = (I == J) ? 16 * I + 4 * lq + rr : 16 * I + li;
                            const int e2 = (I == J) ? 16 * J + li : 16 * J + 4 * lq + rr;
                            const float h = -ht[rr];
                            if (I != J || e1 >= e2) {
                                P.dbg_H[(int64_t)e1 * npadr + e2] = h;
                                P.dbg_H[(int64_t)e2 * npadr + e1] = h;
                            }
                        }
                    }
                }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = v * 64 + lane;
                if (e < npadr) {
                    P.dbg_vec[e] = gv[v];
                    P.dbg_vec[npadr + e] = lo[v];
                    P.dbg_vec[2 * npadr + e] = hi[v];
                }
            }
            if (lane == 0) {
                P.dbg_vec[480] = (float)n;
                P.dbg_vec[481] = (float)npadr;
            }
        }

        STAMP(3);
        // ---------------- interior-point iterations ----------------
        f32x4 Tt[NTILES], Wd[NB];   // register-resident factor
        int status = 1, nit = 0;
        bool first = true;
        // reference point of the gradient: grad(d) = gref + H32 (d - dref); starts at d = 0 with the fp32 g and
        // is replaced ONCE by the float64 structured gradient when the iterate is close (mu < mu_refine)
        double gref[NV];
        float dref[NV];
        bool refined = !(C.mu_refine > 0.0);
        float mu_last = 3.0e38f;
        double* const sbuf = reinterpret_cast<double*>(P.hscratch + (int64_t)blockIdx.x * P.tile_words);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            gref[v] = (double)gv[v];
            dref[v] = 0.f;
        }
        const float inv2n = 1.0f / (float)(2 * n);
        // EARLY ACTIVE-SET POLISH (FTMPC_F32_MU_POLISH > 0): once mu falls below it the bounds with z > s are taken as active and the
        // problem on that set is solved by two multiplier steps with the penalty FTMPC_F32_PW0 max diag(H) on the active bounds -- the
        // Newton matrix's own shape (Sigma = the penalty on the active variables, 0 elsewhere), so a round of the polish IS a pass of
        // this loop: one factorisation, two solves.  Signs verified (an active bound with a negative multiplier leaves, a violated
        // inactive one enters), at most three rounds; the interior-point iterate is kept in the global slot and taken up again if
        // the set does not settle.  (oracle/qp_oracle.py:polish_general is the
        // same method on general rows; scripts/polish_box_study.py the fp32 study: 9.7 -> 8.0 passes, 1.6e-6 f_max from the exact
        // solution against 9e-6 for the iteration run to mu 1e-11.)
        int pol = 0;
        bool pol_tried = !(FTMPC_F32_MU_POLISH > 0.f);
        bool pol_grad = false;   // the float64 gradient has been taken inside this polish
        unsigned pact = 0u;      // bit 2 v: the lower bound of this lane's variable v is active, bit 2 v + 1: the upper one
        float pw = 0.f;
        float* const bkp = P.hscratch + (int64_t)blockIdx.x * P.tile_words + slot_backup_off_words(N);
        for (int it = 0; it <= C.max_iters; ++it) {
            wave_lds_fence();
            const int lane = lane_now();
            const int li = lane & 15, lq = lane >> 4;
            (void)li; (void)lq;
            const bool do_ref = __builtin_amdgcn_readfirstlane(!refined && mu_last < (float)C.mu_refine && (pol_tried || !FTMPC_F32_REFINE_AT_POLISH));
            float dcur[NV];
            if (it == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v) dcur[v] = valid[v] ? ((sl[v] < su[v]) ? lo[v] + sl[v] : hi[v] - su[v]) : 0.f;
            }
            // one accurate (float64, structured) gradient at the current iterate
            auto refresh = [&]() {
                float dnow[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    dnow[v] = valid[v] ? ((sl[v] < su[v]) ? lo[v] + sl[v] : hi[v] - su[v]) : 0.f;
                    const int e = v * 64 + lane;
                    if (e < npadr) xvp[e] = dnow[v];
                }
                wave_lds_fence();
                constexpr int SAVAIL = (NPAD + SH::SEXTRA) * 4;       // bytes of LDS behind dvp
                double* const gout = sbuf;                               // global, n doubles
                if ((N + 1) * 72 <= SAVAIL)
                    struct_grad(C, (glb_cf64*)recg, (lds_f64*)reinterpret_cast<double*>(recbuf), (lds_cf32*)s_Da, (lds_cf32*)xvp,
                                (lds_f64*)reinterpret_cast<double*>(dvp), (glb_f64*)gout, na, lane);
                else
                    struct_grad(C, (glb_cf64*)recg, (lds_f64*)reinterpret_cast<double*>(recbuf), (lds_cf32*)s_Da, (lds_cf32*)xvp,
                                (glb_f64*)(sbuf + slot_stage_off(N)), (glb_f64*)gout, na, lane);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int e = v * 64 + lane;
                    gref[v] = (valid[v] && e < n) ? gout[e] + 2.0 * C.rho * ((double)ubar[v] + (double)dnow[v]) : 0.0;
                    dref[v] = dnow[v];
                    grad[v] = valid[v] ? (float)gref[v] : 0.f;
                }
                refined = true;
                STAMP(8);
            };
            if (do_ref) {
                refresh();
            } else if (it == 0) {
            // gradient at the start point, gref + H (d - dref).  Later iterates do not need the product
            // again: the Newton system just solved gives  H dd = rhs - Sigma dd,  so the gradient follows
            // the step (see the update at the end of the iteration).  For variables at a bound the two
            // terms are large and cancel poorly, but there an error of the gradient only shifts the
            // multiplier; for free variables both terms vanish with the step.
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = v * 64 + lane;
                if (e < npadr) dvp[e] = dcur[v] - dref[v];
            }
            wave_lds_fence();
            {
                // Tile (I,J), I >= J, sits in LDS as -(H_IJ)' in register order: lane (q, col) holds
                // -H[16I + col][16J + 4q + r], r = 0..3.  One read serves both triangles:
                //   y_I[col]   += sum_r H.. d_J[4q+r]   (per-lane dot, then over the row-groups:  row vector)
                //   y_J[4q+r]  += H.. d_I[col]          (then over the 16 columns:  column tile; I > J only)
                // fp32 is enough: the product only spans the distance to the reference point of the gradient.
                f32x2 arow[NB];
                f32x4 acol[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    arow[I] = f32x2{0.f, 0.f};
                    acol[I] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int I = 0; I < NB; ++I) {
                    const float dI = dvp[16 * I + li];
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        const f32x4 t4 = htiles.ld((I * (I + 1)) / 2 + J, lane);
                        const f32x4 d4 = lds4(dvp + 16 * J + 4 * lq);
                        arow[I] += f32x2{t4.x, t4.y} * f32x2{d4.x, d4.y};
                        arow[I] += f32x2{t4.z, t4.w} * f32x2{d4.z, d4.w};
                        if (J < I) acol[J] += t4 * dI;
                    }
                }
                // column-tile parts to natural order through LDS (xvp is free here), row parts by lane select
                wave_lds_fence();
#pragma unroll
                for (int J = 0; J < NB; ++J) {
                    float c0 = acol[J].x, c1 = acol[J].y, c2 = acol[J].z, c3 = acol[J].w;
                    row_sum16x4(c0, c1, c2, c3);
                    const f32x4 c4 = {c0, c1, c2, c3};
                    if (li == 0) *reinterpret_cast<f32x4*>(xvp + 16 * J + 4 * lq) = c4;
                }
                wave_lds_fence();
                float yrow[NB];
#pragma unroll
                for (int I = 0; I < NB; ++I) yrow[I] = quad_sum(arow[I].x + arow[I].y);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * v + j < NB) t = (lq == j) ? yrow[4 * v + j] : t;
                    const int e = v * 64 + lane;
                    const float hd = -(t + ((e < npadr) ? xvp[e] : 0.f));      // the tiles hold -H
                    grad[v] = valid[v] ? (float)((double)hd + gref[v]) : 0.f;
                }
                wave_lds_fence();
            }
            }
            STAMP(4);
            // complementarity
            float t = 0.f;
            if (first) {
                float gm = 0.f, wm = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (valid[v]) {
                        gm = fmaxf(gm, fabsf(grad[v]));
                        wm = fmaxf(wm, ubv[v]);
                    }
                gm = wave_max(gm);
                wm = wave_max(wm);
                const float mu0 = fmaxf(0.02f * gm * wm, 1e-3f);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    zl[v] = valid[v] ? mu0 / sl[v] : 0.f;
                    zu[v] = valid[v] ? mu0 / su[v] : 0.f;
                }
                first = false;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (valid[v]) t += sl[v] * zl[v] + su[v] * zu[v];
            const float mu = wave_sum(t) * inv2n;
            mu_last = mu;
            if (__builtin_amdgcn_readfirstlane(!pol && !(mu >= mu_stop))) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            if (__builtin_amdgcn_readfirstlane(!pol && !pol_tried && mu < FTMPC_F32_MU_POLISH)) {
                pol_tried = true;
                pol = 1;
                float hs = 0.f;
#pragma unroll
                for (int I = 0; I < NB; ++I) {      // max diag(H): lane (q, col) of the diagonal tile holds -H[16 I + col][16 I + 4 q + r]
                    const f32x4 t4 = htiles.ld((I * (I + 1)) / 2 + I, lane);
                    const int r = li - 4 * lq;
                    const float dg = (r == 0) ? t4.x : (r == 1) ? t4.y : (r == 2) ? t4.z : t4.w;
                    if (r >= 0 && r < 4 && 16 * I + li < n) hs = fmaxf(hs, -dg);
                }
                pw = FTMPC_F32_PW0 * wave_max(hs);
                pact = 0u;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    bkp[(0 * NV + v) * 64 + lane] = sl[v];
                    bkp[(1 * NV + v) * 64 + lane] = su[v];
                    bkp[(2 * NV + v) * 64 + lane] = zl[v];
                    bkp[(3 * NV + v) * 64 + lane] = zu[v];
                    bkp[(4 * NV + v) * 64 + lane] = grad[v];
                    const bool al = valid[v] && zl[v] > sl[v], au = valid[v] && zu[v] > su[v];
                    pact |= (al ? 1u : 0u) << (2 * v) | (au ? 2u : 0u) << (2 * v);
                    zl[v] = al ? zl[v] : 0.f;      // the multipliers of the inactive bounds: 0
                    zu[v] = au ? zu[v] : 0.f;
                }
            }
            // back to the interior-point iterate (the polish did not settle, or its matrix did not factorise)
            auto pol_abandon = [&]() {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    sl[v] = bkp[(0 * NV + v) * 64 + lane];
                    su[v] = bkp[(1 * NV + v) * 64 + lane];
                    zl[v] = bkp[(2 * NV + v) * 64 + lane];
                    zu[v] = bkp[(3 * NV + v) * 64 + lane];
                    grad[v] = bkp[(4 * NV + v) * 64 + lane];
                }
                pol = 0;
                refined = false;      // (the gradient kept aside follows the recurrence: the float64 one is taken at the top of the next pass)
            };
            // right-hand side of a multiplier step:  (H + Sigma_A) dd = -grad + C_A' (W s_A - lam),  lower row c = -e, upper row c = +e
            auto pol_rhs = [&](int v) -> float {
                float r = 0.f;
                if (valid[v]) {
                    r = -grad[v];
                    if (pact >> (2 * v) & 1u) r -= pw * sl[v] - zl[v];
                    if (pact >> (2 * v + 1) & 1u) r += pw * su[v] - zu[v];
                }
                return r;
            };
            // the signs on the set: an active bound whose multiplier came out negative leaves, an inactive bound that is violated enters
            auto pol_signs = [&]() -> bool {
                bool changed = false;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (valid[v]) {
                        const unsigned bl = 1u << (2 * v), bu = 2u << (2 * v);
                        if (pact & bl) {
                            if (zl[v] < 0.f) { zl[v] = 0.f; pact &= ~bl; changed = true; }
                        } else if (sl[v] < -1e-6f * ubv[v]) { pact |= bl; changed = true; }
                        if (pact & bu) {
                            if (zu[v] < 0.f) { zu[v] = 0.f; pact &= ~bu; changed = true; }
                        } else if (su[v] < -1e-6f * ubv[v]) { pact |= bu; changed = true; }
                    }
                return __any(changed);
            };
            // ... and its step: the gradient follows by the Newton identity, the multipliers by lam += W (c'dd - s)
            auto pol_step = [&](int v, float r, float dd) {
                if (valid[v]) {
                    const bool al = pact >> (2 * v) & 1u, au = pact >> (2 * v + 1) & 1u;
                    grad[v] += r - ((al ? pw : 0.f) + (au ? pw : 0.f)) * dd;
                    if (al) zl[v] += pw * (-dd - sl[v]);
                    if (au) zu[v] += pw * (dd - su[v]);
                    sl[v] += dd;
                    su[v] -= dd;
                }
            };
            // block column 0 of the Hessian from the global slot (NB > 8, and NB = 8 at two waves per SIMD): requested here, the
            // barrier weights below cover part of the trip
            f32x4 pre0[NB];
            chol_prefetch_col0<NB>(htiles, lane, pre0);
            // KKT matrix: H + Sigma on the diagonal
            float Sig[NV], rsl[NV], rsu[NV];   // 1/s_l, 1/s_u by v_rcp_f32 (1 ulp; the IPM tolerates it)
            wave_lds_fence();   // dvp is dead: it becomes the Sigma vector
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                rsl[v] = __builtin_amdgcn_rcpf(sl[v]);
                rsu[v] = __builtin_amdgcn_rcpf(su[v]);
                Sig[v] = valid[v] ? zl[v] * rsl[v] + zu[v] * rsu[v] : 0.f;
                if (pol) Sig[v] = ((pact >> (2 * v) & 1u) ? pw : 0.f) + ((pact >> (2 * v + 1) & 1u) ? pw : 0.f);
                const int e = v * 64 + lane;
                if (e < npadr) dvp[e] = Sig[v];
            }
            wave_lds_fence();
            STAMP(7);
            const bool ok = chol_reg<NB, TileStore<NLDS>, true, OCC2>(htiles, dvp, recbuf, n, lane, Tt, Wd, pre0, wlds);
            STAMP(5);
            if (__builtin_amdgcn_readfirstlane(!ok)) {
                if (pol) {
                    pol_abandon();
                    continue;
                }
                status = 2;
                break;
            }
            // predictor: (H+Sig) da = -grad   [polish: the first multiplier step]
            float rcl[NV], rcu[NV], rhs[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = v * 64 + lane;
                rhs[v] = pol ? pol_rhs(v) : -grad[v];
                if (e < npadr) xvp[e] = rhs[v];
            }
            wave_lds_fence();
            STAMP(7);
            solve_reg<NB, OCC2>(Tt, Wd, xvp, nbr, lane, wlds);
            STAMP(6);
            float da[NV], dzl_a[NV], dzu_a[NV];
            float ap = 1.f, ad = 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = v * 64 + lane;
                da[v] = (e < npadr && valid[v]) ? xvp[e] : 0.f;
                dzl_a[v] = dzu_a[v] = 0.f;
                if (pol) {
                    pol_step(v, rhs[v], da[v]);
                } else if (valid[v]) {
                    dzl_a[v] = -zl[v] - zl[v] * da[v] * rsl[v];
                    dzu_a[v] = -zu[v] + zu[v] * da[v] * rsu[v];
                    const float rda = __builtin_amdgcn_rcpf(da[v]);
                    if (da[v] < 0.f) ap = fminf(ap, -sl[v] * rda);
                    if (da[v] > 0.f) ap = fminf(ap, su[v] * rda);
                    if (dzl_a[v] < 0.f) ad = fminf(ad, -zl[v] * __builtin_amdgcn_rcpf(dzl_a[v]));
                    if (dzu_a[v] < 0.f) ad = fminf(ad, -zu[v] * __builtin_amdgcn_rcpf(dzu_a[v]));
                }
            }
            // polish, first round: the ONE float64 gradient of this instance is taken HERE, between the two multiplier steps -- the first
            // step (on the gradient the recurrence carried from the start point) lands next to the solution of the active set, the
            // second, from the exact gradient there, is then a true refinement step: its own error scales with its (tiny) length, where
            // a step taken from the interior-point iterate at mu 1e-5 leaves kappa eps |dd| in the flat directions (1.4e-4 f_max measured)
            // The register-resident factor is parked in the global slot around the call (the off-diagonal tiles; the inverse diagonal
            // blocks too where they are not in LDS): left live across it, the allocator spilled it piecemeal through the whole loop (12.2 -> 21.5 ms).
            // The signs are checked after the FIRST step already: a set that fails them goes straight to the next round (new
            // factorisation) without the gradient and the second step, which would be thrown away with it -- so the float64 gradient is
            // taken once per instance, in the round whose set survives its first step (38 % of the instances need a second round).
            // A later round takes it again when its first step was long: that step moves the free variables by what the change of the
            // set is worth, and the recurrence would hide what the fp32 solve leaves of it (9e-5 f_max, worst of 160 000, without).
            bool regrad = false;
            if (pol) {
                if (pol_signs()) {
                    if (++pol > 3) pol_abandon();
                    continue;
                }
                regrad = !pol_grad;
                if (!regrad) {
                    float ddm = 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                        if (valid[v]) ddm = fmaxf(ddm, fabsf(da[v]) * __builtin_amdgcn_rcpf(ubv[v]));
                    regrad = wave_max(ddm) > FTMPC_F32_REGRAD_DD;
                }
                pol_grad = true;
            }
            if (regrad) {
                typedef __attribute__((address_space(1))) f32x4 glb_f32x4;
                glb_f32x4* const fpark = (glb_f32x4*)(P.hscratch + (int64_t)blockIdx.x * P.tile_words + slot_factor_off_words(N));
                int slotn = 0;
#pragma unroll
                for (int I = 1; I < NB; ++I)
#pragma unroll
                    for (int J = 0; J < I; ++J) fpark[(slotn++) * 64 + lane] = Tt[tidx(I, J)];
                if constexpr (!OCC2) {      // (NB = 8 at two waves keeps the inverse diagonal blocks in LDS)
#pragma unroll
                    for (int J = 0; J < NB; ++J) fpark[(slotn++) * 64 + lane] = Wd[J];
                }
                refresh();
                slotn = 0;
#pragma unroll
                for (int I = 1; I < NB; ++I)
#pragma unroll
                    for (int J = 0; J < I; ++J) Tt[tidx(I, J)] = fpark[(slotn++) * 64 + lane];
                if constexpr (!OCC2) {
#pragma unroll
                    for (int J = 0; J < NB; ++J) Wd[J] = fpark[(slotn++) * 64 + lane];
                }
            }
            ap = wave_min(ap);
            ad = wave_min(ad);
            t = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (valid[v]) t += (sl[v] + ap * da[v]) * (zl[v] + ad * dzl_a[v]) + (su[v] - ap * da[v]) * (zu[v] + ad * dzu_a[v]);
            const float mu_aff = wave_sum(t) * inv2n;
            float sigma = mu_aff / mu;
            sigma = fminf(fmaxf(sigma * sigma * sigma, 0.f), 1.f);
            // corrector   [polish: the second multiplier step]
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                rcl[v] = rcu[v] = rhs[v] = 0.f;
                if (pol) {
                    rhs[v] = pol_rhs(v);
                } else if (valid[v]) {
                    rcl[v] = sl[v] * zl[v] + da[v] * dzl_a[v] - sigma * mu;
                    rcu[v] = su[v] * zu[v] - da[v] * dzu_a[v] - sigma * mu;
                    rhs[v] = -(grad[v] - zl[v] + zu[v]) - rcl[v] * rsl[v] + rcu[v] * rsu[v];
                }
                const int e = v * 64 + lane;
                if (e < npadr) xvp[e] = rhs[v];
            }
            wave_lds_fence();
            STAMP(7);
            solve_reg<NB, OCC2>(Tt, Wd, xvp, nbr, lane, wlds);
            STAMP(6);
            float dd[NV], dzl[NV], dzu[NV];
            ap = 1e30f;
            ad = 1e30f;
            if (pol) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int e = v * 64 + lane;
                    pol_step(v, rhs[v], (e < npadr && valid[v]) ? xvp[e] : 0.f);
                }
                const bool changed = pol_signs();
                STAMP(7);
                if (!changed) {
                    status = 0;
                    break;
                }
                if (++pol > 3) pol_abandon();
                continue;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = v * 64 + lane;
                dd[v] = (e < npadr && valid[v]) ? xvp[e] : 0.f;
                dzl[v] = dzu[v] = 0.f;
                if (valid[v]) {
                    dzl[v] = (-rcl[v] - zl[v] * dd[v]) * rsl[v];
                    dzu[v] = (-rcu[v] + zu[v] * dd[v]) * rsu[v];
                    const float rdd = __builtin_amdgcn_rcpf(dd[v]);
                    if (dd[v] < 0.f) ap = fminf(ap, -sl[v] * rdd);
                    if (dd[v] > 0.f) ap = fminf(ap, su[v] * rdd);
                    if (dzl[v] < 0.f) ad = fminf(ad, -zl[v] * __builtin_amdgcn_rcpf(dzl[v]));
                    if (dzu[v] < 0.f) ad = fminf(ad, -zu[v] * __builtin_amdgcn_rcpf(dzu[v]));
                }
            }
            ap = fminf(1.f, 0.9995f * wave_min(ap));
            ad = fminf(1.f, 0.9995f * wave_min(ad));
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (valid[v]) {
                    grad[v] += ap * (rhs[v] - Sig[v] * dd[v]);   // + ap H dd
                    sl[v] += ap * dd[v];
                    su[v] -= ap * dd[v];
                    zl[v] += ad * dzl[v];
                    zu[v] += ad * dzu[v];
                }
            STAMP(7);
        }

        // ---------------- outputs ----------------
        wave_lds_fence();
        float* ubuf = tiles;  // N*NT words, zero = broken thruster
        for (int i = lane; i < N * NT; i += 64) ubuf[i] = 0.f;
        wave_lds_fence();
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (valid[v]) {
                float u = fminf(fmaxf((sl[v] < su[v]) ? sl[v] : ubv[v] - su[v], 0.f), ubv[v]);      // (a polished active bound sits at 0 +- rounding)
                if (status == 2) u = ubar[v];
                ubuf[kcol[v] * NT + s_act[acol[v]]] = u;
            }
        wave_lds_fence();
        if (lane < NT) P.out_u0[inst * NT + lane] = (double)ubuf[lane];
        if (P.out_U)
            for (int i = lane; i < N * NT; i += 64) P.out_U[inst * (int64_t)N * NT + i] = (double)ubuf[i];
        if (lane == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        STAMP(9);
#ifdef FTMPC_STAMPS
        if (lane == 0 && inst < 4096) {
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(P.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = st_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_f32_kernel<8>(const DeviceConsts, const SolveParams);
template __global__ void ftmpc_solve_f32_kernel<9>(const DeviceConsts, const SolveParams);
template __global__ void ftmpc_solve_f32_kernel<10>(const DeviceConsts, const SolveParams);

}  // namespace ftmpc
