// cost.hip -- the correlation cost-matrix build for gfx950 (MI355X), hand-written HIP.
//
// Replaces, on the device, the reference's numpy passes
//   normalize_data              /root/reference/cytospace/common/common.py:142-147
//   matrix_correlation_pearson  /root/reference/cytospace/common/common.py:190-199
//   calculate_cost (lapjv/Pearson branch: negate + repeat each spot row slots[s] times)
//                               /root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:42-69
//
// Formulation.  The reference computes (v2^T v1 - outer(sum2,sum1)/G) / outer(std2,std1) / G in
// float64, i.e. a dot product minus a product of sums (catastrophic cancellation in float32).
// Here every column is first STANDARDISED in float64 and stored as float32
//     z[g][c] = (y[g][c] - mean_c) / (std_c * sqrt(G)),     y = log2(x * 1e6 / colsum + 1)
// so the contraction   corr[s][c] = sum_g zst[g][s] * zsc[g][c]   IS the correlation and needs no
// cancellation; it runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) with fp32
// accumulation.  Zero-variance columns give inf/NaN exactly where the reference divides by zero;
// the LAP entry point then reports CYTO_ERR_NONFINITE.
//
// Kernels
//   colsum_partial_v / colmoments_partial_v / col_finish : per-column statistics (HBM-bound streaming;
//       lanes run along the contiguous cell/spot axis, partial sums over blocks of genes).
//   transform_write_v : z in float32 into a zero-padded [Gpad][pitch] buffer (pitch % 128 == 0)
//   normalize_write   : y in float64 (only for callers that want normalize_data itself)
//   pearson_gemm      : 128x128x32 LDS-tiled TN GEMM on 32x32x2 fp32 MFMA, epilogue negates and
//                       writes each spot row to its slots[s] consecutive LAP rows.
#include "cyto_common.h"
#include <math.h>
#include <algorithm>
#include <chrono>
#include <vector>
#include <thread>
#include <new>
#include <string.h>

namespace cyto {

constexpr int GB = 64;  // genes per partial-statistics block

template <typename TIn> __device__ __forceinline__ double clean(TIn x) {
    // np.nan_to_num: NaN -> 0, +-inf -> +-max finite
    double d = (double)x;
    if (d != d) return 0.0;
    if (d > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (d < -1.7976931348623157e308) return -1.7976931348623157e308;
    return d;
}

// y = log2(x * 1e6 / colsum + 1) with NaN -> 0 (an all-zero column is 0/0)
template <typename TIn> __device__ __forceinline__ double normalized(TIn x, double scale, bool already) {
    if (already) return clean<TIn>(x);
    double y = log2(clean<TIn>(x) * scale + 1.0);
    if (y != y) y = 0.0;
    if (y > 1.7976931348623157e308) y = 1.7976931348623157e308;
    return y;
}

// log2 of a double in [1, 2^1023) in ~25 instructions (ocml's log2: ~120, and the transform evaluates it twice per element -- at 65 f64
// operations each the two passes were bound by the vector ALUs, not by HBM): t = 2^e m, m in [1, 2); the top six mantissa bits select
// c_i = (129 + 2 i) / 128 (c_0 = 1: values next to 1 -- a count of zero is t = 1 exactly -- keep full RELATIVE accuracy, log2(1) = 0),
// r = (m - c_i) / c_i with |r| <= 1 / 64 and m - c_i exact, log2 t = (e + log2 c_i) + log2(1 + r), the last as its Taylor polynomial of
// degree 9 (the first dropped term: 1e-19).  Against long-double log2 on [1, 1e7] and next to 1: relative error <= 2.2e-16 (one ulp).
// tbl[i] = { 1 / c_i, log2 c_i } (a copy of LOG2_TBL in LDS).
__device__ const double LOG2_TBL[64][2] = {
    {1.0, 0.0}, {0.9770992366412213, 0.03342300153745028},
    {0.9624060150375939, 0.0552824355011896}, {0.9481481481481482, 0.0768155970508309},
    {0.9343065693430657, 0.09803208296052672}, {0.920863309352518, 0.11894107272350743},
    {0.9078014184397163, 0.13955135239879354}, {0.8951048951048951, 0.1598713367783894},
    {0.8827586206896552, 0.17990909001493446}, {0.8707482993197279, 0.1996723448363644},
    {0.8590604026845637, 0.21916852046216156}, {0.847682119205298, 0.2384047393250789},
    {0.8366013071895425, 0.25738784269265175}, {0.8258064516129032, 0.27612440527423754},
    {0.8152866242038217, 0.294620748891627}, {0.8050314465408805, 0.31288295528435534},
    {0.7950310559006211, 0.33091687811461695}, {0.7852760736196319, 0.34872815423107756},
    {0.7757575757575758, 0.3663222142458158}, {0.7664670658682635, 0.38370429247405224},
    {0.757396449704142, 0.4008794362821843}, {0.7485380116959064, 0.41785251488589786},
    {0.7398843930635838, 0.43462822763672465}, {0.7314285714285714, 0.4512111118323288},
    {0.7231638418079096, 0.4676055500829974}, {0.7150837988826816, 0.4838157772642564},
    {0.7071823204419889, 0.4998458870832054}, {0.6994535519125683, 0.5156998382840424},
    {0.6918918918918919, 0.5313814605163121}, {0.6844919786096256, 0.5468944598876366},
    {0.6772486772486772, 0.5622424242210726}, {0.6701570680628273, 0.5774288280357487},
    {0.6632124352331606, 0.5924570372680804}, {0.6564102564102564, 0.6073303137496107},
    {0.649746192893401, 0.6220518194563762}, {0.6432160804020101, 0.6366246205436489},
    {0.6368159203980099, 0.6510516911789286}, {0.6305418719211823, 0.6653359171851763},
    {0.624390243902439, 0.6794800995054461}, {0.6183574879227053, 0.6934869574993252},
    {0.6124401913875598, 0.7073591320808827}, {0.6066350710900474, 0.7210991887071851},
    {0.6009389671361502, 0.7347096202258382}, {0.5953488372093023, 0.7481928495894603},
    {0.5898617511520737, 0.7615512324444793}, {0.5844748858447488, 0.7747870596011734},
    {0.579185520361991, 0.7879025593914316}, {0.5739910313901345, 0.8008998999203047},
    {0.5688888888888889, 0.8137811912170371}, {0.5638766519823789, 0.826548487290915},
    {0.5589519650655022, 0.839203788096944}, {0.5541125541125541, 0.8517490414160576},
    {0.5493562231759657, 0.8641861446542802}, {0.5446808510638298, 0.8765169465649997},
    {0.540084388185654, 0.8887432488982591}, {0.5355648535564853, 0.9008668079807486},
    {0.5311203319502075, 0.9128893362299616}, {0.5267489711934157, 0.9248125036057809},
    {0.5224489795918368, 0.9366379390025705}, {0.5182186234817814, 0.9483672315846776},
    {0.5140562248995983, 0.9600019320680809}, {0.5099601593625498, 0.971543553950772},
    {0.5059288537549407, 0.9829935746943101}, {0.5019607843137255, 0.9943534368588579}};
__device__ __forceinline__ double log2_ge1(double t, const double2 *__restrict__ tbl) {
    const int hi = __double2hiint(t), lo = __double2loint(t);
    const int e = ((hi >> 20) & 0x7FF) - 1023;
    const double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, lo);
    const int i = (hi >> 14) & 63;
    const double2 ck = tbl[i];
    const double c = i ? (double)(129 + 2 * i) * 0.0078125 : 1.0;
    const double r = (m - c) * ck.x;
    double p = 0.1602994489876626;
    p = fma(p, r, -0.18033688011112042);
    p = fma(p, r, 0.2060992915555662);
    p = fma(p, r, -0.2404491734814939);
    p = fma(p, r, 0.28853900817779266);
    p = fma(p, r, -0.36067376022224085);
    p = fma(p, r, 0.4808983469629878);
    p = fma(p, r, -0.7213475204444817);
    p = fma(p, r, 1.4426950408889634);
    return fma(r, p, (double)e + ck.y);
}
// (out of line: the rare path's ~120 instructions stay out of the streaming loops)
__device__ __attribute__((noinline)) double log2_cleaned(double t) {
    double y = log2(t);
    if (y != y) y = 0.0;
    if (y > 1.7976931348623157e308) y = 1.7976931348623157e308;
    return y;
}
// `normalized` with the fast logarithm wherever its argument is an ordinary number >= 1 (every non-negative count of a column with a
// positive sum); anything else -- negative or non-finite input, an all-zero column's 0 * inf -- takes the library's log2 as before
template <typename TIn, int VEC>
__device__ __forceinline__ void normalized_f(const TIn (&x)[VEC], const double (&scale)[VEC], bool already, const double2 *__restrict__ tbl,
                                             double (&y)[VEC]) {
    if (already) {
#pragma unroll
        for (int e = 0; e < VEC; e++) y[e] = clean<TIn>(x[e]);
        return;
    }
    double t[VEC];
    bool all_ok = true;
#pragma unroll
    for (int e = 0; e < VEC; e++) {
        t[e] = clean<TIn>(x[e]) * scale[e] + 1.0;
        const bool ok = t[e] >= 1.0 && t[e] < 8.0e307;
        all_ok = all_ok && ok;
        y[e] = log2_ge1(ok ? t[e] : 1.0, tbl);                   // (straight-line for every lane; the odd ones are redone below)
    }
    if (!all_ok) {
#pragma unroll
        for (int e = 0; e < VEC; e++)
            if (!(t[e] >= 1.0 && t[e] < 8.0e307)) y[e] = log2_cleaned(t[e]);
    }
}
__device__ __forceinline__ void load_log2_table(double2 *tbl) {      // (64 entries; the caller synchronises)
    if (threadIdx.x < 64) tbl[threadIdx.x] = make_double2(LOG2_TBL[threadIdx.x][0], LOG2_TBL[threadIdx.x][1]);
}

// ---- the transform's streaming kernels (round 4) -------------------------------------------------------------------------------
// A block = four waves on the same 64 * VEC columns; wave rs takes the genes g0 + rs, g0 + rs + 4, ... of a block of TGB genes, a lane
// VEC consecutive columns (one 16-byte load per row for float32 counts, eight rows in flight), the four waves' partial sums are
// combined in wave order through LDS: a column's sum is the same whatever VEC is (VEC = 1: any pitch / alignment).  The forms above
// (a thread per column, scalar loads, 313 partials per column at G = 20 000) ran the three passes + a full memset of z at 1.8 TB/s.
constexpr int TGB = 256;   // genes per block
constexpr int TRS = 4;     // waves per block = row subgroups

template <typename TIn, int VEC> __device__ __forceinline__ void load_cols(const TIn *__restrict__ p, TIn (&r)[VEC]) {
    if constexpr (VEC == 1) {
        r[0] = p[0];
    } else if constexpr (sizeof(TIn) == 2) {                        // uint16 counts: four columns = one 8-byte load
        const ushort4 t = *reinterpret_cast<const ushort4 *>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (sizeof(TIn) == 1) {                        // uint8 counts: one 4-byte load
        const uchar4 t = *reinterpret_cast<const uchar4 *>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (sizeof(TIn) == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
        const double2 a = reinterpret_cast<const double2 *>(p)[0], b = reinterpret_cast<const double2 *>(p)[1];
        r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
    }
}

// combine the four waves' partials (wave order) and store them as block blockIdx.y's partial
template <int VEC>
__device__ __forceinline__ void block_combine_store(double (&s)[VEC], double (*sh)[64 * VEC], int lane, int rs, bool act,
                                                    double *__restrict__ part, int C, int c0) {
    if (rs > 0) {
#pragma unroll
        for (int e = 0; e < VEC; e++) sh[rs - 1][e * 64 + lane] = s[e];
    }
    __syncthreads();
    if (rs == 0 && act) {
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            double t = s[e];
#pragma unroll
            for (int k = 0; k < TRS - 1; k++) t += sh[k][e * 64 + lane];
            part[(int64_t)blockIdx.y * C + c0 + e] = t;
        }
    }
}

template <typename TIn, int VEC>
__global__ __launch_bounds__(64 * TRS) void colsum_partial_v(int G, int C, const TIn *__restrict__ x, int64_t ldx,
                                                             double *__restrict__ part) {
    __shared__ double sh[TRS - 1][64 * VEC];
    const int lane = threadIdx.x & 63, rs = threadIdx.x >> 6;
    const int c0 = ((int)blockIdx.x * 64 + lane) * VEC;
    const bool act = c0 < C;
    const int g0 = blockIdx.y * TGB, g1 = min(G, g0 + TGB);
    double s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) s[e] = 0.0;
    if (act) {
        const TIn *__restrict__ p = x + (int64_t)(g0 + rs) * ldx + c0;
#pragma unroll 8
        for (int g = g0 + rs; g < g1; g += TRS, p += TRS * ldx) {
            TIn r[VEC];
            load_cols<TIn, VEC>(p, r);
#pragma unroll
            for (int e = 0; e < VEC; e++) s[e] += clean<TIn>(r[e]);
        }
    }
    block_combine_store<VEC>(s, sh, lane, rs, act, part, C, c0);
}

template <typename TIn, int VEC>
__global__ __launch_bounds__(64 * TRS) void colmoments_partial_v(int G, int C, const TIn *__restrict__ x, int64_t ldx,
                                                                 const double *__restrict__ colsum, int already,
                                                                 double *__restrict__ p1, double *__restrict__ p2) {
    __shared__ double sh[TRS - 1][64 * VEC];
    __shared__ double2 tbl[64];
    load_log2_table(tbl);
    __syncthreads();
    const int lane = threadIdx.x & 63, rs = threadIdx.x >> 6;
    const int c0 = ((int)blockIdx.x * 64 + lane) * VEC;
    const bool act = c0 < C;
    const int g0 = blockIdx.y * TGB, g1 = min(G, g0 + TGB);
    double s1[VEC], s2[VEC], scale[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) { s1[e] = 0.0; s2[e] = 0.0; scale[e] = (already || !act) ? 1.0 : 1e6 / colsum[c0 + e]; }
    if (act) {
        const TIn *__restrict__ p = x + (int64_t)(g0 + rs) * ldx + c0;
#pragma unroll 4
        for (int g = g0 + rs; g < g1; g += TRS, p += TRS * ldx) {
            TIn r[VEC];
            load_cols<TIn, VEC>(p, r);
            double y[VEC];
            normalized_f<TIn, VEC>(r, scale, already, tbl, y);
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                s1[e] += y[e];
                s2[e] += y[e] * y[e];
            }
        }
    }
    block_combine_store<VEC>(s1, sh, lane, rs, act, p1, C, c0);
    __syncthreads();                                              // (the staging lines are read before the second moments overwrite them)
    block_combine_store<VEC>(s2, sh, lane, rs, act, p2, C, c0);
}

// z = (y - mean) * inv (MODE 0) or z = y (MODE 1: the Euclidean metric's operand) as float32
template <typename TIn, int VEC, int MODE>
__global__ __launch_bounds__(64 * TRS) void transform_write_v(int G, int C, const TIn *__restrict__ x, int64_t ldx,
                                                              const double *__restrict__ colsum, const double *__restrict__ mean,
                                                              const double *__restrict__ inv, int already,
                                                              float *__restrict__ z, int64_t ldz) {
    __shared__ double2 tbl[64];
    load_log2_table(tbl);
    __syncthreads();
    const int lane = threadIdx.x & 63, rs = threadIdx.x >> 6;
    const int c0 = ((int)blockIdx.x * 64 + lane) * VEC;
    if (c0 >= C) return;
    const int g0 = blockIdx.y * TGB, g1 = min(G, g0 + TGB);
    double scale[VEC], m[VEC], iv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) {
        scale[e] = already ? 1.0 : 1e6 / colsum[c0 + e];
        m[e] = MODE == 0 ? mean[c0 + e] : 0.0;
        iv[e] = MODE == 0 ? inv[c0 + e] : 1.0;
    }
    const TIn *__restrict__ p = x + (int64_t)(g0 + rs) * ldx + c0;
    float *__restrict__ q = z + (int64_t)(g0 + rs) * ldz + c0;
#pragma unroll 4
    for (int g = g0 + rs; g < g1; g += TRS, p += TRS * ldx, q += TRS * ldz) {
        TIn r[VEC];
        load_cols<TIn, VEC>(p, r);
        float o[VEC];
        double y[VEC];
        normalized_f<TIn, VEC>(r, scale, already, tbl, y);
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = MODE == 0 ? (float)((y[e] - m[e]) * iv[e]) : (float)y[e];
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(q) = make_float4(o[0], o[1], o[2], o[3]);
        else q[0] = o[0];
    }
}

// combine the partials in ascending block order
__global__ void col_finish_sum(int C, int nblk, const double *__restrict__ part, double *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; b++) s += part[(int64_t)b * C + c];
    out[c] = s;
}

// mean and 1 / (std * sqrt(G)) per column (population std, ddof = 0, as numpy's default)
__global__ void col_finish_moments(int G, int C, int nblk, const double *__restrict__ p1, const double *__restrict__ p2,
                                   double *__restrict__ mean, double *__restrict__ inv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; b++) { s1 += p1[(int64_t)b * C + c]; s2 += p2[(int64_t)b * C + c]; }
    const double m = s1 / G;
    double var = s2 / G - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = m;
    inv[c] = 1.0 / (sqrt(var) * sqrt((double)G));   // var == 0 -> inf, like the reference's /0
}

template <typename TIn>
__global__ __launch_bounds__(256) void normalize_write(int G, int C, const TIn *__restrict__ x, int64_t ldx,
                                                       const double *__restrict__ colsum, double *__restrict__ y, int64_t ldy) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int g0 = blockIdx.y * GB, g1 = min(G, g0 + GB);
    const double scale = 1e6 / colsum[c];
    for (int g = g0; g < g1; g++) y[(int64_t)g * ldy + c] = normalized<TIn>(x[(int64_t)g * ldx + c], scale, false);
}

// partial column sums of squares of a float32 matrix (the rounded values the matrix cores will see)
__global__ __launch_bounds__(256) void colsq_partial(int G, int C, const float *__restrict__ z, int64_t ldz, double *__restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int g0 = blockIdx.y * GB, g1 = min(G, g0 + GB);
    double s = 0.0;
    for (int g = g0; g < g1; g++) { const double v = (double)z[(int64_t)g * ldz + c]; s += v * v; }
    part[(int64_t)blockIdx.y * C + c] = s;
}

// Per-column ranks, ties averaged, 1-based: pandas.DataFrame.rank() defaults, as used by
// matrix_correlation_spearman (/root/reference/cytospace/common/common.py:202-215).
// rank_i = #{x_j < x_i} + (#{x_j == x_i} + 1) / 2, exact in float64 order: one workgroup per column; the column is
// taken in chunks of 8192 values that are sorted in LDS as order-preserving 64-bit keys (bitonic network, 1024
// threads), and every value of the column binary-searches each sorted chunk for its lower and upper bound.
// Ranks are half-integers <= G: exact in float32.  The transform normalize_data applies before
// (log2(x * 1e6 / colsum + 1)) is strictly increasing per column, so ranking the cleaned input gives the ranks of
// the normalised values.
constexpr int RANK_CHUNK = 8192;
__device__ __forceinline__ uint64_t d2ord(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x + 0.0);          // +0.0: -0 -> +0
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
// A launch ranks the values own0 <= g < own0 + MOWN * 1024 of every column (against the WHOLE column): unfiltered 10x gene
// sets (33 538 or 36 601 genes) take two launches.
template <typename TIn, int MOWN>
__global__ __launch_bounds__(1024) void rank_columns(int G, int C, const TIn *__restrict__ x, int64_t ldx,
                                                     float *__restrict__ r, int64_t ldr, int own0) {
    __shared__ uint64_t sk[RANK_CHUNK];
    const int c = blockIdx.x, tid = threadIdx.x;
    uint64_t own[MOWN];
    int lt[MOWN], eq[MOWN];
#pragma unroll
    for (int e = 0; e < MOWN; e++) {
        const int g = own0 + tid + e * 1024;
        own[e] = g < G ? d2ord(clean<TIn>(x[(int64_t)g * ldx + c])) : 0ull;
        lt[e] = 0; eq[e] = 0;
    }
    for (int j0 = 0; j0 < G; j0 += RANK_CHUNK) {
        __syncthreads();
        for (int j = tid; j < RANK_CHUNK; j += 1024)
            sk[j] = (j0 + j < G) ? d2ord(clean<TIn>(x[(int64_t)(j0 + j) * ldx + c])) : ~0ull;   // pad: above every key
        __syncthreads();
        for (int k = 2; k <= RANK_CHUNK; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < RANK_CHUNK / 2; t += 1024) {
                    const int i = 2 * t - (t & (j - 1)), l = i + j;
                    const uint64_t a = sk[i], b2 = sk[l];
                    if ((a > b2) == ((i & k) == 0)) { sk[i] = b2; sk[l] = a; }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int e = 0; e < MOWN; e++) {
            if (own0 + tid + e * 1024 < G) {
                const uint64_t key = own[e];
                int lo = 0, hi = 0;
#pragma unroll
                for (int step = RANK_CHUNK / 2; step >= 1; step >>= 1) {
                    if (sk[lo + step - 1] < key) lo += step;
                    if (sk[hi + step - 1] <= key) hi += step;
                }
                if (sk[lo] < key) lo++;
                if (sk[hi] <= key) hi++;
                lt[e] += lo; eq[e] += hi - lo;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < MOWN; e++) {
        const int g = own0 + tid + e * 1024;
        if (g < G) r[(int64_t)g * ldr + c] = (float)((double)lt[e] + 0.5 * (double)(eq[e] + 1));
    }
}

// ------------------------------------------------------------------------------------------
// cost[r][c] = - sum_g zst[g][s] * zsc[g][c]   for every LAP row r of spot s
// TN GEMM: both operands are stored gene-major, so a k-slice of either tile is a contiguous row
// segment (coalesced 16-byte loads straight into LDS, conflict-free fragment reads).
// Block tile 128 (spots) x 128 (cells) x 32 (genes), 4 waves, each wave 64x64 = 2x2 MFMA tiles, two workgroups per CU.
//
// WHAT BOUNDS THE TILE LOOP (tools/dbg/mfma_peak.hip): the matrix pipe sustains 99 % of its peak with this MFMA sequence and
// these LDS fragment reads, but every VALU instruction issued among the MFMAs costs it 4-8 cycles.  The first version of
// this kernel spent 34 LDS-address adds, 32 packed adds (fold), ~16 global-address computations and a burst of eight
// ds_write_b128 per 32-gene tile and reached 79 % (124 TFLOP/s at c3 size).  Now the loop needs (almost) no VALU work:
// LDS fragments by ds_read2st64_b32 (one loop-invariant per-lane base + immediates), operands by buffer loads straight
// into LDS (scalar row offsets), the fold every second tile -- 140 TFLOP/s = 89 %.
// ------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;
#ifndef GEMM_FOLD
#define GEMM_FOLD 2
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

// EPI 0: cost = -dot (Pearson / Spearman on standardised operands);
// EPI 1: cost = sqrt(|a|^2 + |b|^2 - 2 dot) (Euclidean; squared norms in float64, sum and root in float64)
template <int EPI>
__global__ __launch_bounds__(256, 2) void pearson_gemm(int Gpad, int S, int C, const float *__restrict__ A, int64_t lda,
                                                    const float *__restrict__ B, int64_t ldb,
                                                    const int *__restrict__ rowstart, float *__restrict__ cost, int64_t ldc,
                                                    int tiles_n, const double *__restrict__ na, const double *__restrict__ nb, int tiles_m) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (private L2s), so every XCD gets a contiguous
    // band of the tile sequence; inside the band tiles are walked in SUPER x SUPER blocks, so the ~64 workgroups an XCD runs
    // at a time share 8 spot panels and 8 cell panels in its L2 (row-major order: 1 spot panel + 64 cell panels, four
    // times the operand traffic per flop).
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int SUPER = 8;
    int tm, tn;
    {
        const int band = SUPER * tiles_n;                 // tiles in a full band of SUPER tile rows
        const int bi = wg / band, rem = wg - bi * band;
        const int rows_here = min(SUPER, tiles_m - bi * SUPER);
        const int cb = rem / (rows_here * SUPER), rem2 = rem - cb * rows_here * SUPER;
        const int cols_here = min(SUPER, tiles_n - cb * SUPER);
        tm = bi * SUPER + rem2 / cols_here;
        tn = cb * SUPER + rem2 % cols_here;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave >> 1, wn = wave & 1;

    // Two-level accumulation: the matrix cores add into `acc` for FOLD k-tiles (32 genes each), then `acc` is folded into
    // `sum` with ordinary round-to-nearest adds and restarted from zero (C = 0 in the next tile's first MFMAs).  The MFMA
    // accumulate truncates: one running fp32 sum over 20 000 genes (config c3) ended ~1e-5 below the float64 reference (up to
    // 7e-5), always towards zero; partial sums of 32 genes are ~600x smaller, so are their truncation steps.  Measured at c3
    // size with FOLD = 1: every entry within 2e-6 (FOLD = 8: 21 of 4.8 M entries above, FOLD = 16: up to 3e-6 on
    // correlations near 1).  The 32 packed adds per fold are not free: every VALU instruction costs the matrix pipe 4-8 cycles.
    constexpr int FOLD = GEMM_FOLD;
    static_assert(FOLD == 1 || FOLD == 2, "the tile loop is unrolled by two");
    f32x16 acc[2][2], sum[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[a][b][r] = 0.0f; sum[a][b][r] = 0.0f; }

    // staging: direct-to-LDS buffer loads (buffer_load_dwordx4 ... lds), double buffered.  The descriptor's base is the
    // (scalar) start of the tile's 32-row slab, the four passes are scalar offsets, the thread's place in the slab is ONE
    // loop-invariant register -- no vector address arithmetic per tile.  Lane l of a wave writes 16 bytes at M0 + 16 l,
    // i.e. a wave fills the two 128-float k-rows 2w + 8t, 2w + 8t + 1 of the tile; no staging registers, no ds_write.
    // (Round 1-2 staged through eight float4 registers and eight ds_write_b128: same speed once the stores were spread over the
    //  tile, but 90 more VGPRs.)
    const int st_k = tid >> 5, st_c = (tid & 31) * 4;   // this thread's row / column inside a staging pass
    const uint32_t voA = (uint32_t)(((int64_t)st_k * lda + m0 + st_c) * 4), voB = (uint32_t)(((int64_t)st_k * ldb + n0 + st_c) * 4);
    const uint32_t passA = (uint32_t)(8 * lda * 4), passB = (uint32_t)(8 * ldb * 4);
    const int nk = Gpad / BK;
    typedef __attribute__((address_space(3))) void lds_void;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#define DL1(Xs, BUF, rs, vo, pass, t) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)&Xs[BUF][2 * wave_s + 8 * (t)][0], 16, vo, (t) * pass, 0, 0);
#define DLOAD(k0, BUF)                                                                              \
    {                                                                                               \
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + (int64_t)(k0) * lda), 0, 0x7FFFFFFF, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B + (int64_t)(k0) * ldb), 0, 0x7FFFFFFF, 0x00020000); \
        DL1(As, BUF, rsA, voA, passA, 0) DL1(As, BUF, rsA, voA, passA, 1) DL1(As, BUF, rsA, voA, passA, 2) DL1(As, BUF, rsA, voA, passA, 3) \
        DL1(Bs, BUF, rsB, voB, passB, 0) DL1(Bs, BUF, rsB, voB, passB, 1) DL1(Bs, BUF, rsB, voB, passB, 2) DL1(Bs, BUF, rsB, voB, passB, 3) \
    }
    // the loads' arrival in LDS is awaited explicitly (vmcnt): the fragment reads below are inline asm the compiler cannot see
#define DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DLOAD(0, 0)
    DMA_WAIT()
    __syncthreads();
    const int li = lane & 31, lk = lane >> 5;
    const int ao = wm * 64 + li, bo = wn * 64 + li;
    // LDS fragment reads as ds_read2st64_b32 (inline asm: the compiler pairs a0/a1 into ds_read2_b32, whose 8-bit offsets
    // need a vector add per read): base = one loop-invariant per-lane register, the two offsets (units of 256 bytes) select
    // the buffer and the k-steps kk and kk + 2.  The waits for these reads are explicit (the compiler does not see them):
    // LDS operations retire in order, so "at most the four reads just issued are outstanding" covers the pair before.
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const uint32_t pa0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&As[0][lk][ao], pa1 = pa0 + 128;
    const uint32_t pb0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&Bs[0][lk][bo], pb1 = pb0 + 128;
#define DSR(dst, base, o0) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(base), "n"(o0), "n"((o0) + 4));
#define FRAGS(BUF, kk, A0, A1, B0, B1) DSR(A0, pa0, (BUF) * 64 + (kk) * 2) DSR(A1, pa1, (BUF) * 64 + (kk) * 2)                   \
                                       DSR(B0, pb0, (BUF) * 64 + (kk) * 2) DSR(B1, pb1, (BUF) * 64 + (kk) * 2)
#define WAITF(n, A0, A1, B0, B1) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1));
#define MFMA4(a0_, a1_, b0_, b1_)                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b0_, acc[0][0], 0, 0, 0);                           \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b1_, acc[0][1], 0, 0, 0);                           \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b0_, acc[1][0], 0, 0, 0);                           \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc[1][1], 0, 0, 0);
    // pair p of a tile: steps kk = 4p (x) and 4p + 2 (y) from the registers of set CUR; the reads of pair p + 1 go to set NXT
#define PAIR(BUF, p, CA0, CA1, CB0, CB1, NA0, NA1, NB0, NB1)                                                  \
    if ((p) < 7) { FRAGS(BUF, 4 * (p) + 4, NA0, NA1, NB0, NB1) }                                              \
    WAITF(4, CA0, CA1, CB0, CB1)                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    MFMA4(CA0.x, CA1.x, CB0.x, CB1.x)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    MFMA4(CA0.y, CA1.y, CB0.y, CB1.y)                                                                         \
    __builtin_amdgcn_sched_barrier(0);
    // One tile out of buffer BUF (compile-time: the loop is unrolled over the two buffers).  On entry set x holds the
    // fragments of its first pair and the direct-to-LDS loads of tile kt + 1 are in flight into the other buffer.  The
    // workgroup barrier sits BEFORE the last pair: by then every read of this buffer has been issued (pair 7's fragments
    // during pair 6) and is complete (the barrier's lgkmcnt(0)), and tile kt + 1 has landed (vmcnt(0)) -- so the next tile's
    // first fragments are requested right after the barrier and arrive behind the eight MFMAs of pair 7, and the loads of
    // tile kt + 2 can already overwrite THIS tile's buffer.
#define TILE(BUF, FOLDNOW)                                                                                    \
    {                                                                                                         \
        const bool more = kt + 1 < nk;                                                                        \
        PAIR(BUF, 0, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1)                                                  \
        PAIR(BUF, 1, ya0, ya1, yb0, yb1, xa0, xa1, xb0, xb1)                                                  \
        PAIR(BUF, 2, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1)                                                  \
        PAIR(BUF, 3, ya0, ya1, yb0, yb1, xa0, xa1, xb0, xb1)                                                  \
        PAIR(BUF, 4, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1)                                                  \
        PAIR(BUF, 5, ya0, ya1, yb0, yb1, xa0, xa1, xb0, xb1)                                                  \
        PAIR(BUF, 6, xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1)                                                  \
        DMA_WAIT()                                                                                            \
        __syncthreads();                                                                                      \
        if (more) { FRAGS((BUF) ^ 1, 0, xa0, xa1, xb0, xb1) }                                                 \
        if (kt + 2 < nk) { DLOAD((kt + 2) * BK, BUF) }                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        PAIR(BUF, 7, ya0, ya1, yb0, yb1, xa0, xa1, xb0, xb1)                                                  \
        if (FOLDNOW || !more) {                                                                               \
            _Pragma("unroll") for (int a = 0; a < 2; a++)                                                     \
                _Pragma("unroll") for (int b = 0; b < 2; b++) {                                               \
                    sum[a][b] += acc[a][b];                                                                   \
                    _Pragma("unroll") for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;                       \
                }                                                                                             \
        }                                                                                                     \
        kt++;                                                                                                 \
    }
    f32x2v xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
    int kt = 0;
    FRAGS(0, 0, xa0, xa1, xb0, xb1)
    if (nk > 1) { DLOAD(BK, 1) }
    while (kt + 1 < nk) {
        TILE(0, FOLD == 1)
        TILE(1, true)
    }
    if (kt < nk) TILE(0, true)
#undef DSR
#undef FRAGS
#undef WAITF
#undef MFMA4
#undef PAIR
#undef TILE
#undef DL1
#undef DLOAD
#undef DMA_WAIT

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // rowstart == nullptr: every slot count is 1, row = spot (the unique-row storage and single-cell mode).  Otherwise the
    // 16 row ranges of an accumulator are fetched together first (one round trip, not sixteen dependent ones).
#pragma unroll
    for (int a = 0; a < 2; a++) {
        int r0s[16], r1s[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int s = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            r0s[r] = s; r1s[r] = s + 1;
            if (rowstart && s < S) { r0s[r] = rowstart[s]; r1s[r] = rowstart[s + 1]; }
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int c = n0 + wn * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int s = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (s < S && c < C) {
                    float val;
                    if constexpr (EPI == 0) val = -sum[a][b][r];
                    else { const double d2 = na[s] + nb[c] - 2.0 * (double)sum[a][b][r]; val = (float)sqrt(d2 > 0.0 ? d2 : 0.0); }
                    if (!rowstart) cost[(int64_t)s * ldc + c] = val;
                    else for (int row = r0s[r]; row < r1s[r]; row++) cost[(int64_t)row * ldc + c] = val;
                }
            }
        }
    }
}

// out[g][c'] = z[g][idx[c']] for c' < nsel (columns beyond nsel are left zero): the GEMM operand of one chunk
__global__ __launch_bounds__(256) void gather_columns(int Gpad, int nsel, const float *__restrict__ z, int64_t ldz,
                                                      const int32_t *__restrict__ idx, float *__restrict__ out, int64_t ldo) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nsel) return;
    const int src = idx[c];
    const int g0 = blockIdx.y * GB, g1 = min(Gpad, g0 + GB);
    for (int g = g0; g < g1; g++) out[(int64_t)g * ldo + c] = z[(int64_t)g * ldz + src];
}

// Sparse counts -> dense: one thread per stored entry (binary search of its column in colptr); the dense matrix was zeroed.
__global__ __launch_bounds__(256) void csc_scatter(int C, int64_t nnz, const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                   const float *__restrict__ vals, float *__restrict__ dense, int64_t ld, int G,
                                                   int *__restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    int lo = 0, hi = C;                           // largest c with colptr[c] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (colptr[mid] <= e) lo = mid; else hi = mid; }
    const int g = rowidx[e];
    if (g < 0 || g >= G) { *bad = 1; return; }
    dense[(int64_t)g * ld + lo] = vals[e];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Column statistics + standardised float32 matrix for one input (genes x columns).
// z must hold Gpad x ldz floats and be zero-filled by the caller (padding must stay zero).
// transform: 0 standardise (Pearson), 1 rank then standardise (Spearman), 2 plain float32 conversion (Euclidean)
template <typename TIn>
static int standardize_dev(int G, int C, const TIn *dx, int64_t ldx, int already, float *z, int64_t ldz, double *ynorm,
                           int64_t ldy, hipStream_t stream, int transform = 0) {
    if (transform == 1) {
        // ranks into a float32 matrix, then the usual standardisation of that matrix
        if (G >= (1 << 24)) return CYTO_ERR_UNSUPPORTED;      // ranks are stored as float32: exact up to 2^24 values per column
        DevBuf ranks;
        int rc2;
        if ((rc2 = ranks.alloc((size_t)G * C * sizeof(float), stream))) return rc2;
        float *rp = ranks.as<float>();
        for (int own0 = 0; own0 < G; own0 += 32 * 1024) {
            const int m = (min(G - own0, 32 * 1024) + 1023) / 1024;
            if (m <= 4) hipLaunchKernelGGL((rank_columns<TIn, 4>), dim3(C), dim3(1024), 0, stream, G, C, dx, ldx, rp, (int64_t)C, own0);
            else if (m <= 8) hipLaunchKernelGGL((rank_columns<TIn, 8>), dim3(C), dim3(1024), 0, stream, G, C, dx, ldx, rp, (int64_t)C, own0);
            else if (m <= 16) hipLaunchKernelGGL((rank_columns<TIn, 16>), dim3(C), dim3(1024), 0, stream, G, C, dx, ldx, rp, (int64_t)C, own0);
            else if (m <= 24) hipLaunchKernelGGL((rank_columns<TIn, 24>), dim3(C), dim3(1024), 0, stream, G, C, dx, ldx, rp, (int64_t)C, own0);
            else hipLaunchKernelGGL((rank_columns<TIn, 32>), dim3(C), dim3(1024), 0, stream, G, C, dx, ldx, rp, (int64_t)C, own0);
        }
        CYTO_HIP(hipGetLastError());
        return standardize_dev<float>(G, C, rp, C, 1, z, ldz, nullptr, 0, stream, 0);
    }
    const bool want_moments = z && transform != 2;
    const int nblk = (G + TGB - 1) / TGB, nblk_y = (G + GB - 1) / GB;
    DevBuf part1, part2, colsum, mean, inv;
    int rc;
    if ((rc = part1.alloc((size_t)nblk * C * sizeof(double), stream)) ||
        (want_moments && (rc = part2.alloc((size_t)nblk * C * sizeof(double), stream))) ||
        (rc = colsum.alloc((size_t)C * sizeof(double), stream)) || (rc = mean.alloc((size_t)C * sizeof(double), stream)) ||
        (rc = inv.alloc((size_t)C * sizeof(double), stream)))
        return rc;
    // four columns per lane when every row of x (and of z) can be read (written) as 16-byte quads; else a column per lane -- a column's
    // numbers do not depend on which
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) % (sizeof(TIn) >= 4 ? 16 : 4 * sizeof(TIn))) == 0 &&
                     (!z || (ldz % 4 == 0 && (reinterpret_cast<uintptr_t>(z) % 16) == 0));
    const int cols_per_block = vec ? 256 : 64;
    const dim3 grid((C + cols_per_block - 1) / cols_per_block, nblk), blk(64 * TRS);
    const dim3 g1((C + 255) / 256), blk1(256);
    if (!already) {
        if (vec) hipLaunchKernelGGL((colsum_partial_v<TIn, 4>), grid, blk, 0, stream, G, C, dx, ldx, part1.as<double>());
        else hipLaunchKernelGGL((colsum_partial_v<TIn, 1>), grid, blk, 0, stream, G, C, dx, ldx, part1.as<double>());
        hipLaunchKernelGGL(col_finish_sum, g1, blk1, 0, stream, C, nblk, part1.as<double>(), colsum.as<double>());
    }
    if (ynorm) {
        hipLaunchKernelGGL(normalize_write<TIn>, dim3((C + 255) / 256, nblk_y), blk1, 0, stream, G, C, dx, ldx, colsum.as<double>(), ynorm, ldy);
    }
    if (z && transform == 2) {
        if (vec) hipLaunchKernelGGL((transform_write_v<TIn, 4, 1>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), nullptr, nullptr, already, z, ldz);
        else hipLaunchKernelGGL((transform_write_v<TIn, 1, 1>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), nullptr, nullptr, already, z, ldz);
    } else if (z) {
        if (vec) hipLaunchKernelGGL((colmoments_partial_v<TIn, 4>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), already,
                                    part1.as<double>(), part2.as<double>());
        else hipLaunchKernelGGL((colmoments_partial_v<TIn, 1>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), already,
                                part1.as<double>(), part2.as<double>());
        hipLaunchKernelGGL(col_finish_moments, g1, blk1, 0, stream, G, C, nblk, part1.as<double>(), part2.as<double>(),
                           mean.as<double>(), inv.as<double>());
        if (vec) hipLaunchKernelGGL((transform_write_v<TIn, 4, 0>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), mean.as<double>(),
                                    inv.as<double>(), already, z, ldz);
        else hipLaunchKernelGGL((transform_write_v<TIn, 1, 0>), grid, blk, 0, stream, G, C, dx, ldx, colsum.as<double>(), mean.as<double>(),
                                inv.as<double>(), already, z, ldz);
    }
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipStreamSynchronize(stream));   // the temporaries above die with this scope
    return CYTO_OK;
}

// The element type of a count matrix at the boundary (cytohip.h: CYTO_DTYPE_*; the parameters still called `x_is_f64` carry it: 0 and 1
// mean what they always meant).  Raw counts are small integers: uint16 / uint8 matrices are a half / a quarter of the float32 upload
// (config c3: 4.4 GB at ~50 GB/s were 93 of 120 ms), widened in the transform kernels' loads -- the same numbers, bit for bit.
static inline size_t dtype_size(int dt) { return dt == CYTO_DTYPE_F64 ? 8 : dt == CYTO_DTYPE_U16 ? 2 : dt == CYTO_DTYPE_U8 ? 1 : 4; }
static inline bool dtype_ok(int dt) { return dt >= CYTO_DTYPE_F32 && dt <= CYTO_DTYPE_U8; }
static int standardize_any(int dt, int G, int C, const void *dx, int64_t ldx, int already, float *z, int64_t ldz, double *ynorm, int64_t ldy,
                           hipStream_t stream, int transform) {
    switch (dt) {
    case CYTO_DTYPE_F64: return standardize_dev<double>(G, C, (const double *)dx, ldx, already, z, ldz, ynorm, ldy, stream, transform);
    case CYTO_DTYPE_U16: return standardize_dev<uint16_t>(G, C, (const uint16_t *)dx, ldx, already, z, ldz, ynorm, ldy, stream, transform);
    case CYTO_DTYPE_U8: return standardize_dev<uint8_t>(G, C, (const uint8_t *)dx, ldx, already, z, ldz, ynorm, ldy, stream, transform);
    case CYTO_DTYPE_F32: return standardize_dev<float>(G, C, (const float *)dx, ldx, already, z, ldz, ynorm, ldy, stream, transform);
    default: return CYTO_ERR_BAD_ARG;
    }
}

}  // namespace cyto

using namespace cyto;

extern "C" {

// A1: normalize_data (common/common.py:142-147) on the device; out is float64 like the reference's.
int cyto_normalize_data(int G, int C, const void *x, int64_t ldx, int x_is_f64, double *out, int64_t ldo, int device_id) {
    if (G <= 0 || C <= 0 || !x || !out || ldx < C || ldo < C) return CYTO_ERR_BAD_ARG;
    int rc = select_device(device_id);
    if (rc) return rc;
    if (!dtype_ok(x_is_f64)) return CYTO_ERR_BAD_ARG;
    const size_t esz = dtype_size(x_is_f64);
    DevBuf dx, dy;
    if ((rc = dx.alloc((size_t)G * C * esz)) || (rc = dy.alloc((size_t)G * C * 8))) return rc;
    CYTO_HIP(hipMemcpy2D(dx.p, (size_t)C * esz, x, (size_t)ldx * esz, (size_t)C * esz, G, hipMemcpyHostToDevice));
    if ((rc = standardize_any(x_is_f64, G, C, dx.p, C, 0, nullptr, 0, dy.as<double>(), C, nullptr, 0))) return rc;
    CYTO_HIP(hipMemcpy2D(out, (size_t)ldo * 8, dy.p, (size_t)C * 8, (size_t)C * 8, G, hipMemcpyDeviceToHost));
    return CYTO_OK;
}

// A1+A2 (first half): per-column normalise (unless already_normalized) and standardise; writes the
// float32 matrix z (device pointer, Gpad x ldz, pre-zeroed by this call) used by cyto_cost_pearson.
int cyto_transform(int transform, int G, int C, const void *x, int64_t ldx, int x_is_f64, int x_on_device, int already_normalized,
                   float *z_dev, int64_t ldz, int Gpad, int device_id, void *stream_) {
    if (G <= 0 || C <= 0 || !x || !z_dev || ldx < C || ldz < C || Gpad < G) return CYTO_ERR_BAD_ARG;
    if (transform < CYTO_TRANSFORM_STANDARDIZE || transform > CYTO_TRANSFORM_RAW || !dtype_ok(x_is_f64)) return CYTO_ERR_BAD_ARG;
    int rc = select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const size_t esz = dtype_size(x_is_f64);
    DevBuf dx;
    const void *src = x;
    int64_t sld = ldx;
    if (!x_on_device) {
        if ((rc = dx.alloc((size_t)G * C * esz, stream))) return rc;
        CYTO_HIP(hipMemcpy2DAsync(dx.p, (size_t)C * esz, x, (size_t)ldx * esz, (size_t)C * esz, G, hipMemcpyHostToDevice, stream));
        src = dx.p;
        sld = C;
    }
    // the zero padding of z: the columns beyond C and the rows beyond G (the transform writes every element of the G x C block itself;
    // a memset of the whole buffer was a fifth pass over a matrix the transform reads three times and writes once)
    if (ldz > C) CYTO_HIP(hipMemset2DAsync(z_dev + C, (size_t)ldz * sizeof(float), 0, (size_t)(ldz - C) * sizeof(float), (size_t)G, stream));
    if (Gpad > G) CYTO_HIP(hipMemsetAsync(z_dev + (size_t)G * ldz, 0, (size_t)(Gpad - G) * ldz * sizeof(float), stream));
    return standardize_any(x_is_f64, G, C, src, sld, already_normalized, z_dev, ldz, nullptr, 0, stream, transform);
}

int cyto_standardize(int G, int C, const void *x, int64_t ldx, int x_is_f64, int x_on_device, int already_normalized,
                     float *z_dev, int64_t ldz, int Gpad, int device_id, void *stream_) {
    return cyto_transform(CYTO_TRANSFORM_STANDARDIZE, G, C, x, ldx, x_is_f64, x_on_device, already_normalized, z_dev, ldz, Gpad,
                          device_id, stream_);
}

// the -corr contraction, launch only (rowstart == nullptr: one cost row per spot)
static void gemm_unique_rows_async(int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                                   const int *rowstart, float *cost_dev, int64_t ldc, hipStream_t stream) {
    const int tiles_m = (S + BM - 1) / BM, tiles_n = (C + BN - 1) / BN;
    hipLaunchKernelGGL(pearson_gemm<0>, dim3(tiles_m * tiles_n), dim3(256), 0, stream, Gpad, S, C, zst, ldzst, zsc, ldzsc,
                       rowstart, cost_dev, ldc, tiles_n, (const double *)nullptr, (const double *)nullptr, tiles_m);
}

// A2+A3: cost = -corr, each spot row written to its slots[s] LAP rows (spot order).
// zst: Gpad x ldzst, zsc: Gpad x ldzsc (device, zero padded: Gpad % 32 == 0, ld % 128 == 0).
// cost: device, (sum slots) x ldc.  gemm_ms (optional): HIP-event time of the GEMM kernel.
static int cost_gemm(int euclid, int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                     const int64_t *slots, float *cost_dev, int64_t ldc, double *gemm_ms, int device_id, void *stream_) {
    if (Gpad <= 0 || S <= 0 || C <= 0 || !zst || !zsc || !slots || !cost_dev) return CYTO_ERR_BAD_ARG;
    if (Gpad % BK || ldzst % BM || ldzsc % BN || ldzst < S || ldzsc < C || ldc < C) return CYTO_ERR_BAD_ARG;
    int rc = select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    std::vector<int> rowstart((size_t)S + 1);
    int64_t acc = 0;
    bool identity = true;                                  // every slot count 1 (unique-row storage, single-cell mode): row = spot
    for (int s = 0; s < S; s++) {
        if (slots[s] < 0) return CYTO_ERR_BAD_ARG;
        rowstart[s] = (int)acc;
        acc += slots[s];
        identity = identity && slots[s] == 1;
        if (acc > 0x7FFFFFFF) return CYTO_ERR_UNSUPPORTED;
    }
    rowstart[S] = (int)acc;
    DevBuf drs;
    if (!identity) {
        if ((rc = drs.alloc(((size_t)S + 1) * sizeof(int), stream))) return rc;
        CYTO_HIP(hipMemcpyAsync(drs.p, rowstart.data(), ((size_t)S + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    const int tiles_m = (S + BM - 1) / BM, tiles_n = (C + BN - 1) / BN;
    Events<2> ev;
    if ((rc = ev.create())) return rc;
    const hipEvent_t e0 = ev[0], e1 = ev[1];
    DevBuf na, nb, part;
    if (euclid) {
        // squared column norms of the float32 operands, in float64
        const int nblk = Gpad / GB + (Gpad % GB ? 1 : 0);
        const int Cmax = S > C ? S : C;
        if ((rc = na.alloc((size_t)S * 8, stream)) || (rc = nb.alloc((size_t)C * 8, stream)) || (rc = part.alloc((size_t)nblk * Cmax * 8, stream))) return rc;
        hipLaunchKernelGGL(colsq_partial, dim3((S + 255) / 256, nblk), dim3(256), 0, stream, Gpad, S, zst, ldzst, part.as<double>());
        hipLaunchKernelGGL(col_finish_sum, dim3((S + 255) / 256), dim3(256), 0, stream, S, nblk, part.as<double>(), na.as<double>());
        hipLaunchKernelGGL(colsq_partial, dim3((C + 255) / 256, nblk), dim3(256), 0, stream, Gpad, C, zsc, ldzsc, part.as<double>());
        hipLaunchKernelGGL(col_finish_sum, dim3((C + 255) / 256), dim3(256), 0, stream, C, nblk, part.as<double>(), nb.as<double>());
    }
    CYTO_HIP(hipEventRecord(e0, stream));
    if (euclid)
        hipLaunchKernelGGL(pearson_gemm<1>, dim3(tiles_m * tiles_n), dim3(256), 0, stream, Gpad, S, C, zst, ldzst, zsc, ldzsc,
                           drs.as<int>(), cost_dev, ldc, tiles_n, na.as<double>(), nb.as<double>(), tiles_m);
    else
        gemm_unique_rows_async(Gpad, S, C, zst, ldzst, zsc, ldzsc, identity ? nullptr : drs.as<int>(), cost_dev, ldc, stream);
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipEventRecord(e1, stream));
    CYTO_HIP(hipStreamSynchronize(stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (gemm_ms) *gemm_ms = ms;
    return CYTO_OK;
}

int cyto_cost_pearson(int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                      const int64_t *slots, float *cost_dev, int64_t ldc, double *gemm_ms, int device_id, void *stream_) {
    return cost_gemm(0, Gpad, S, C, zst, ldzst, zsc, ldzsc, slots, cost_dev, ldc, gemm_ms, device_id, stream_);
}

// The other distance metrics of calculate_cost (linear_assignment_solvers.py:53-59) through the same contraction:
// Spearman = the Pearson epilogue on rank-transformed operands (cyto_transform(CYTO_TRANSFORM_RANK, ...));
// Euclidean = sqrt(|a|^2 + |b|^2 - 2 a.b) on the plain float32 operands (cyto_transform(CYTO_TRANSFORM_RAW, ...)).
int cyto_cost_metric(int metric, int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                     const int64_t *slots, float *cost_dev, int64_t ldc, double *gemm_ms, int device_id, void *stream_) {
    if (metric < CYTO_METRIC_PEARSON || metric > CYTO_METRIC_EUCLIDEAN) return CYTO_ERR_BAD_ARG;
    return cost_gemm(metric == CYTO_METRIC_EUCLIDEAN, Gpad, S, C, zst, ldzst, zsc, ldzsc, slots, cost_dev, ldc, gemm_ms, device_id, stream_);
}

// A7: the fused per-chunk path of solve_linear_assignment_problem (cytospace/cytospace.py:304-351)
// for solver_method == "lapjv", distance_metric == "Pearson_correlation": cost build on the device,
// JV solve on the device, mapped_spot[c] = spot of the LAP row given to cell c.
// sc: G x C, st: G x S (host, row-major, float64 like the reference's arrays).  sum(slots) must be C.
// The 1e-16 perturbation of cytospace.py:325-327 is not applied: it is a no-op in float32.
static int assign_metric_impl(int metric, int G, int C, int S, const void *sc, int sc_dt, const void *st, int st_dt, const int64_t *slots,
                              int already_normalized, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id) {
    if (metric < CYTO_METRIC_PEARSON || metric > CYTO_METRIC_EUCLIDEAN || !dtype_ok(sc_dt) || !dtype_ok(st_dt)) return CYTO_ERR_BAD_ARG;
    const int transform = metric == CYTO_METRIC_PEARSON ? CYTO_TRANSFORM_STANDARDIZE
                        : metric == CYTO_METRIC_SPEARMAN ? CYTO_TRANSFORM_RANK : CYTO_TRANSFORM_RAW;
    if (G <= 0 || C <= 0 || S <= 0 || !sc || !st || !slots || !mapped_spot) return CYTO_ERR_BAD_ARG;
    int64_t N = 0;
    for (int s = 0; s < S; s++) { if (slots[s] < 0) return CYTO_ERR_BAD_ARG; N += slots[s]; }
    if (N != C) return CYTO_ERR_BAD_ARG;   // the LAP must be square
    int rc = select_device(device_id);
    if (rc) return rc;
    const int Gpad = (int)round_up(G, BK);
    const int64_t ldzst = round_up(S, BM), ldzsc = round_up(C, BN), ldc = round_up(C, 4);
    // a private stream, so that several host threads can run chunks concurrently on one GPU
    // (declared before the buffers: they go back to the block cache while their stream still exists)
    StreamGuard guard;
    if ((rc = guard.acquire())) return rc;
    const hipStream_t stream = guard.s;
    Events<2> ev;
    if ((rc = ev.create())) return rc;
    const hipEvent_t e0 = ev[0], e1 = ev[1];
    // Row indirection: the cost holds ONE row per spot (S x C); LAP row i reads the row of its spot through rowspot[]
    // (calculate_cost materialises the repeats: linear_assignment_solvers.py:63-66 -- 10 GB instead of 1 GB at config c3).
    std::vector<int32_t> rowspot((size_t)N);
    {
        int64_t r = 0;
        for (int s = 0; s < S; s++) for (int64_t k = 0; k < slots[s]; k++) rowspot[(size_t)r++] = s;
    }
    std::vector<int64_t> ones((size_t)S, 1);
    DevBuf zst, zsc, cost;
    if ((rc = zst.alloc((size_t)Gpad * ldzst * 4, stream)) || (rc = zsc.alloc((size_t)Gpad * ldzsc * 4, stream)) ||
        (rc = cost.alloc((size_t)S * ldc * 4, stream)))
        return rc;
    CYTO_HIP(hipEventRecord(e0, stream));
    if ((rc = cyto_transform(transform, G, S, st, S, st_dt, 0, already_normalized, zst.as<float>(), ldzst, Gpad, device_id, stream))) return rc;
    float ms_std = 0;
    double ms_gemm = 0;
    constexpr int WBLK = 8192;                            // cells per block: 64 column tiles x 40 row tiles at c3 = five full rounds
    if (metric != CYTO_METRIC_EUCLIDEAN && C > 2 * WBLK) {
        // Large problems: the scRNA matrix goes up in blocks of cells, and the contraction of block b (its own stream) runs
        // while block b + 1 is copied and transformed -- every transform is per cell (column), and a pitched copy out of
        // pageable memory runs at the same 57 GB/s as a contiguous one (tools/dbg/h2d_2d_bench.hip).  At c3 the upload
        // (70 ms) and the contraction (71 ms) then overlap instead of adding up.
        // (Copying on a third stream with the transforms enqueued without a host wait was measured too: no faster -- while a
        //  contraction runs the upload itself slows down from 57 to ~40 GB/s.)
        StreamGuard gcomp;
        if ((rc = gcomp.acquire())) return rc;
        const size_t esz = dtype_size(sc_dt);
        DevBuf dx;
        if ((rc = dx.alloc((size_t)G * WBLK * esz, stream))) return rc;
        // (only the padding of the operand is cleared -- the columns beyond C, the genes beyond G: the transforms write every element of
        //  the G x C block themselves, and a memset of the whole 4-GB buffer was another pass in front of the first block)
        if (ldzsc > C) CYTO_HIP(hipMemset2DAsync(zsc.as<float>() + C, (size_t)ldzsc * sizeof(float), 0, (size_t)(ldzsc - C) * sizeof(float), (size_t)G, stream));
        if (Gpad > G) CYTO_HIP(hipMemsetAsync(zsc.as<float>() + (size_t)G * ldzsc, 0, (size_t)(Gpad - G) * ldzsc * sizeof(float), stream));
        // the FIRST block is a quarter of the others: nothing is contracted before it has crossed PCIe and been transformed, so the
        // pipeline starts 2 048 cells in instead of 8 192 (the block's own contraction fills the chip 1.25 times: 4 % of the work)
        std::vector<int> bstart;
        for (int c0 = 0; c0 < C; c0 += (c0 == 0 ? WBLK / 4 : WBLK)) bstart.push_back(c0);
        bstart.push_back(C);
        const int nblocks = (int)bstart.size() - 1;
        std::vector<hipEvent_t> evs((size_t)2 * nblocks, nullptr);     // per block: contraction begin / end
        struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } evguard{evs};
        for (size_t k = 0; k < evs.size(); k++) CYTO_HIP(hipEventCreate(&evs[k]));
        // every early return below leaves contractions queued on gcomp.s that write `cost` and read zst / zsc: both streams drain
        // before any of the buffers is released (the block cache may hand a released block to another thread's solve at once)
        StreamDrain drain_comp{gcomp.s}, drain_main{stream};
        for (int b = 0; b < nblocks; b++) {
            const int c0 = bstart[(size_t)b], w = bstart[(size_t)b + 1] - c0;
            const char *src = reinterpret_cast<const char *>(sc) + (size_t)c0 * esz;
            CYTO_HIP(hipMemcpy2DAsync(dx.p, (size_t)w * esz, src, (size_t)C * esz, (size_t)w * esz, G, hipMemcpyHostToDevice, stream));
            rc = standardize_any(sc_dt, G, w, dx.p, w, already_normalized, zsc.as<float>() + c0, ldzsc, nullptr, 0, stream, transform);
            if (rc) return rc;                            // (standardize_dev has waited for its kernels: block b's operand is complete)
            CYTO_HIP(hipEventRecord(evs[2 * b], gcomp.s));
            gemm_unique_rows_async(Gpad, S, w, zst.as<float>(), ldzst, zsc.as<float>() + c0, ldzsc, nullptr, cost.as<float>() + c0, ldc, gcomp.s);
            CYTO_HIP(hipGetLastError());
            CYTO_HIP(hipEventRecord(evs[2 * b + 1], gcomp.s));
        }
        CYTO_HIP(hipEventRecord(e1, stream));
        CYTO_HIP(hipStreamSynchronize(gcomp.s));
        CYTO_HIP(hipEventSynchronize(e1));
        (void)hipEventElapsedTime(&ms_std, e0, e1);     // upload + transforms; the contractions of all blocks but the last ran inside it
        for (int b = 0; b < nblocks; b++) { float ms = 0; (void)hipEventElapsedTime(&ms, evs[2 * b], evs[2 * b + 1]); ms_gemm += ms; }
    } else {
        if ((rc = cyto_transform(transform, G, C, sc, C, sc_dt, 0, already_normalized, zsc.as<float>(), ldzsc, Gpad, device_id, stream))) return rc;
        CYTO_HIP(hipEventRecord(e1, stream));
        CYTO_HIP(hipEventSynchronize(e1));
        (void)hipEventElapsedTime(&ms_std, e0, e1);
        if ((rc = cyto_cost_metric(metric, Gpad, S, C, zst.as<float>(), ldzst, zsc.as<float>(), ldzsc, ones.data(), cost.as<float>(), ldc,
                                   &ms_gemm, device_id, stream)))
            return rc;
    }
    std::vector<int32_t> colsol((size_t)N);
    cyto_lap_info li;
    double total = 0;
    bool identity = N == S;                              // every spot takes exactly one cell: row i IS spot i
    for (int s = 0; s < S && identity; s++) identity = slots[s] == 1;
    rc = identity ? cyto_lap_f32((int)N, cost.as<float>(), ldc, 1, nullptr, colsol.data(), nullptr, nullptr, &total, &li, device_id, stream)
                  : cyto_lap_f32_rowmap((int)N, cost.as<float>(), ldc, S, 1, rowspot.data(), nullptr, colsol.data(), nullptr, nullptr, &total,
                                        &li, device_id, stream, nullptr);
    if (rc) return rc;
    // location_repeat[y] (cytospace.py:331): LAP row -> spot
    for (int64_t c = 0; c < C; c++) mapped_spot[c] = rowspot[(size_t)colsol[(size_t)c]];
    if (total_cost) *total_cost = total;
    if (info) {
        memset(info, 0, sizeof *info);
        info->ms_standardize = ms_std;   // includes the host-to-device copies of sc and st
        info->ms_gemm = ms_gemm;
        info->lap = li;
        info->gemm_flops = 2.0 * Gpad * (double)S * (double)C;
    }
    return CYTO_OK;
}

int cyto_assign_metric_typed(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64, const int64_t *slots,
                             int already_normalized, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id) {
    return assign_metric_impl(metric, G, C, S, sc, x_is_f64, st, x_is_f64, slots, already_normalized, mapped_spot, total_cost, info, device_id);
}

// the same with a dtype per matrix (host matrices, row-major, ld == columns): counts as uint8 / uint16 where they fit
int cyto_assign_metric_ex(int metric, int G, const cyto_matrix *sc, int C, const cyto_matrix *st, int S, const int64_t *slots,
                          int already_normalized, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id) {
    if (!sc || !st || sc->on_device || st->on_device || sc->ld != C || st->ld != S) return CYTO_ERR_BAD_ARG;
    return assign_metric_impl(metric, G, C, S, sc->data, sc->is_f64, st->data, st->is_f64, slots, already_normalized, mapped_spot, total_cost, info,
                              device_id);
}

// ---- multi-chunk seam (apply_linear_assignment, cytospace/cytospace.py:354-469): the two expression matrices are
// uploaded and transformed ONCE; every chunk then gathers its cells / spots out of the resident operands ----
struct cyto_expr_ctx {
    int metric, G, Gpad, C, S, device_id;
    int64_t ldsc, ldst;
    DevBuf zsc, zst;
};

// comm == NULL: this process transforms both matrices itself.  comm != NULL (one process per GPU): only `root` holds the
// ST matrix; it transforms it once and the float32 operand goes to the other ranks with ONE ncclBroadcast over xGMI -- the
// only collective of the path (the reference pickles the whole ST matrix to every worker: cytospace.py:438, 446-451).
// sc always holds THIS rank's cells only.
static int ctx_create_impl(int metric, int G, int C, int S, const void *sc, int64_t ldsc_in, int sc_is_f64, int sc_on_device,
                           const void *st, int64_t ldst_in, int st_is_f64, int st_on_device, int already_normalized,
                           void *comm, int root, int rank, int device_id, cyto_expr_ctx **out, double *bcast_ms) {
    // With a communicator every rank MUST reach every collective below whatever went wrong on it -- a rank that returned early
    // would leave the others blocked inside RCCL.  So errors are collected in `rc` and NOTHING returns between here and the last
    // collective; the ranks then agree in two small steps before any data moves:
    //   1. the root's status, raw extents and metric travel first (4 words): a rank whose G / S / metric are not the root's fails
    //      with CYTO_ERR_BAD_ARG (raw, not padded, extents: two different G inside one padding bucket would scale by different
    //      1/sqrt(G) around the same operand);
    //   2. a max-allreduce of "did anything fail here" (local validation, device selection, allocations, the transforms): only
    //      if NO rank failed does anybody enter the operand broadcast -- all ranks enter it or none does, a failed rank needs no
    //      scratch buffer to receive into, and every rank of a failed call returns an error (its own, else the root's, else
    //      CYTO_ERR_PEER).
    // The small collectives go through the communicator's own pre-allocated device word (comm.hip): nothing on the way into
    // them can fail.
    int rc = CYTO_OK;
    if (!out) return CYTO_ERR_BAD_ARG;                                       // (a caller bug on this rank alone: nothing to agree on)
    *out = nullptr;
    if (G <= 0 || C <= 0 || S <= 0 || !sc || ldsc_in < C) rc = CYTO_ERR_BAD_ARG;
    if (metric < CYTO_METRIC_PEARSON || metric > CYTO_METRIC_EUCLIDEAN) rc = CYTO_ERR_BAD_ARG;
    if (comm && (root < 0 || rank != comm_rank(comm))) rc = CYTO_ERR_BAD_ARG;
    if (comm && device_id != comm_device(comm)) { rc = CYTO_ERR_BAD_ARG; device_id = comm_device(comm); }     // (the collectives run on the communicator's device)
    const bool have_st = !comm || rank == root;
    if (have_st && (!st || ldst_in < S)) rc = CYTO_ERR_BAD_ARG;
    if (!comm && rc) return rc;
    const int dev_rc = select_device(device_id);
    if (dev_rc) { if (!comm) return dev_rc; rc = rc ? rc : dev_rc; }
    cyto_expr_ctx *ctx = rc ? nullptr : new (std::nothrow) cyto_expr_ctx();
    if (!rc && !ctx) rc = CYTO_ERR_NOMEM;
    const int transform = metric == CYTO_METRIC_PEARSON ? CYTO_TRANSFORM_STANDARDIZE
                        : metric == CYTO_METRIC_SPEARMAN ? CYTO_TRANSFORM_RANK : CYTO_TRANSFORM_RAW;
    size_t nst = 0;
    if (!rc) {
        ctx->metric = metric; ctx->G = G; ctx->Gpad = (int)round_up(G, BK); ctx->C = C; ctx->S = S; ctx->device_id = device_id;
        ctx->ldsc = round_up(C, BN); ctx->ldst = round_up(S, BM);
        nst = (size_t)ctx->Gpad * ctx->ldst;
        if (!(rc = ctx->zsc.alloc((size_t)ctx->Gpad * ctx->ldsc * 4))) rc = ctx->zst.alloc(nst * 4);
    }
    auto transform_sc = [&]() {
        if (!rc) rc = cyto_transform(transform, G, C, sc, ldsc_in, sc_is_f64, sc_on_device, already_normalized, ctx->zsc.as<float>(), ctx->ldsc,
                                     ctx->Gpad, device_id, nullptr);
    };
    // ranks without the ST matrix transform their own cells FIRST: that work overlaps the root's ST transform instead of
    // waiting behind the broadcast
    if (comm && rank != root) transform_sc();
    if (!rc && have_st)
        rc = cyto_transform(transform, G, S, st, ldst_in, st_is_f64, st_on_device, already_normalized, ctx->zst.as<float>(), ctx->ldst,
                            ctx->Gpad, device_id, nullptr);
    if (comm) {
        int32_t words[4] = {rc, G, S, metric};                               // (only the root's values are sent)
        const int brc = comm_bcast_words(comm, words, 4, root);
        const int root_rc = brc ? 0 : words[0];
        if (!rc && !brc && !root_rc && (words[1] != G || words[2] != S || words[3] != metric)) rc = CYTO_ERR_BAD_ARG;   // not the root's problem
        int any = (rc || brc || root_rc) ? 1 : 0;
        const int arc = comm_allreduce_max(comm, &any);
        if (!rc) rc = brc ? brc : arc;
        if (!rc && root_rc) rc = root_rc;                                    // the root failed: every rank reports the root's code
        if (!rc && any) rc = CYTO_ERR_PEER;                                  // some other rank failed
        if (!rc) {                                                           // (`any` is the same on every rank: all enter, or none)
            Events<2> ev;
            const bool timed = ev.create() == CYTO_OK;
            if (timed) (void)hipEventRecord(ev[0], nullptr);
            rc = comm_bcast_dev(comm, ctx->zst.p, nst * 4, root, nullptr);
            if (timed) {
                (void)hipEventRecord(ev[1], nullptr);
                float ms = 0;
                if (hipEventSynchronize(ev[1]) == hipSuccess && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess && bcast_ms) *bcast_ms = ms;
            }
        }
    }
    if (!comm || rank == root) transform_sc();
    if (rc) { delete ctx; return rc; }
    *out = ctx;
    return CYTO_OK;
}

int cyto_ctx_create_typed(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64, int already_normalized,
                          int device_id, cyto_expr_ctx **out) {
    if (!st) return CYTO_ERR_BAD_ARG;
    return ctx_create_impl(metric, G, C, S, sc, C, x_is_f64, 0, st, S, x_is_f64, 0, already_normalized, nullptr, 0, 0, device_id, out, nullptr);
}

int cyto_ctx_create_shared(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64, int already_normalized,
                           void *comm, int root, int rank, int device_id, cyto_expr_ctx **out, double *bcast_ms) {
    if (!comm) return CYTO_ERR_BAD_ARG;
    return ctx_create_impl(metric, G, C, S, sc, C, x_is_f64, 0, st, S, x_is_f64, 0, already_normalized, comm, root, rank, device_id, out, bcast_ms);
}

int cyto_ctx_create_ex(int metric, int G, const cyto_matrix *sc, int C, const cyto_matrix *st, int S, int already_normalized,
                       void *comm, int root, int rank, int device_id, cyto_expr_ctx **out, double *bcast_ms) {
    if (!sc) return CYTO_ERR_BAD_ARG;
    return ctx_create_impl(metric, G, C, S, sc->data, sc->ld, sc->is_f64, sc->on_device, st ? st->data : nullptr, st ? st->ld : 0,
                           st ? st->is_f64 : 0, st ? st->on_device : 0, already_normalized, comm, root, rank, device_id, out, bcast_ms);
}

// sparse counts (what scipy.io.mmread of a 10x matrix.mtx holds, common/common.py:49) go to the device AS
// non-zeros and are expanded there; the reference densifies on the host (common.py:57) and ships the dense matrix.
int cyto_csc_to_dense_f32(int G, int C, int64_t nnz, const int64_t *colptr, const int32_t *rowidx, const float *vals,
                          float *dense_dev, int64_t ld, int device_id, void *stream_) {
    if (G <= 0 || C <= 0 || nnz < 0 || !colptr || (nnz > 0 && (!rowidx || !vals)) || !dense_dev || ld < C) return CYTO_ERR_BAD_ARG;
    if (colptr[0] != 0 || colptr[C] != nnz) return CYTO_ERR_BAD_ARG;
    for (int c = 0; c < C; c++) if (colptr[c + 1] < colptr[c]) return CYTO_ERR_BAD_ARG;
    int rc = select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    DevBuf dptr, didx, dval, dbad;
    if ((rc = dptr.alloc(((size_t)C + 1) * 8, stream)) || (rc = didx.alloc((size_t)nnz * 4, stream)) || (rc = dval.alloc((size_t)nnz * 4, stream)) ||
        (rc = dbad.alloc(4, stream)))
        return rc;
    CYTO_HIP(hipMemcpyAsync(dptr.p, colptr, ((size_t)C + 1) * 8, hipMemcpyHostToDevice, stream));
    if (nnz) {
        CYTO_HIP(hipMemcpyAsync(didx.p, rowidx, (size_t)nnz * 4, hipMemcpyHostToDevice, stream));
        CYTO_HIP(hipMemcpyAsync(dval.p, vals, (size_t)nnz * 4, hipMemcpyHostToDevice, stream));
    }
    CYTO_HIP(hipMemsetAsync(dbad.p, 0, 4, stream));
    CYTO_HIP(hipMemsetAsync(dense_dev, 0, (size_t)G * ld * 4, stream));
    if (nnz)
        hipLaunchKernelGGL(csc_scatter, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, C, nnz, dptr.as<int64_t>(), didx.as<int32_t>(),
                           dval.as<float>(), dense_dev, ld, G, dbad.as<int>());
    CYTO_HIP(hipGetLastError());
    int bad = 0;
    CYTO_HIP(hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, stream));
    CYTO_HIP(hipStreamSynchronize(stream));
    return bad ? CYTO_ERR_BAD_ARG : CYTO_OK;
}

void cyto_ctx_destroy(cyto_expr_ctx *ctx) {
    if (!ctx) return;
    (void)select_device(ctx->device_id);
    delete ctx;
}

// One chunk: cells idx_sc[0..n_sc) against spots idx_st[0..n_st) (NULL: all S spots) with slots[k] cells for the k-th
// listed spot; sum(slots) must be n_sc.  mapped_spot[c] = position in the chunk's spot list (what the reference's
// solve_linear_assignment_problem returns for the chunk, cytospace.py:331-332).  Spots with slots == 0 are not
// contracted at all.  Thread-safe on one context (private stream and buffers per call).
int cyto_ctx_assign_chunks(cyto_expr_ctx *ctx, int nchunks, cyto_chunk *chunks, int max_concurrent);
int cyto_ctx_assign_chunk(cyto_expr_ctx *ctx, const int64_t *idx_sc, int n_sc, const int64_t *idx_st, int n_st,
                          const int64_t *slots, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info) {
    if (!ctx || !idx_sc || n_sc <= 0 || !slots || !mapped_spot) return CYTO_ERR_BAD_ARG;
    cyto_chunk ch;
    memset(&ch, 0, sizeof ch);
    ch.idx_sc = idx_sc; ch.n_sc = n_sc; ch.idx_st = idx_st; ch.n_st = n_st; ch.slots = slots; ch.mapped_spot = mapped_spot;
    const int rc = cyto_ctx_assign_chunks(ctx, 1, &ch, 1);       // a batch of one (thread-safe: private stream and buffers per call)
    if (total_cost) *total_cost = ch.total_cost;
    if (info) *info = ch.info;
    return rc ? rc : ch.status;
}

// Every chunk of this rank in ONE call: per chunk the two column gathers and the cost GEMM (back to back on one stream,
// the gathered operands are reused), then the LAPs of all chunks TOGETHER -- one workgroup per chunk in every chain phase
// (cyto_lap_batch_f32), so a rank's chunks run on as many CUs side by side.  Replaces the per-chunk worker processes of
// apply_linear_assignment (cytospace/cytospace.py:430-467).  chunks[k].status reports per-chunk failures.
int cyto_ctx_assign_chunks(cyto_expr_ctx *ctx, int nchunks, cyto_chunk *chunks, int max_concurrent) {
    if (!ctx || nchunks < 0 || (nchunks > 0 && !chunks)) return CYTO_ERR_BAD_ARG;
    if (nchunks == 0) return CYTO_OK;
    int rc = select_device(ctx->device_id);
    if (rc) return rc;
    // (developer knob CYTO_TRACE_CHUNKS: wall-clock stamps of this call's steps on stderr -- tools/c3_walls.py)
    const bool trace = CYTO_KNOB("CYTO_TRACE_CHUNKS").set && CYTO_KNOB("CYTO_TRACE_CHUNKS").value;
    const auto t_entry = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (trace) fprintf(stderr, "[chunks] %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(), what);
    };
    const int conc = std::max(1, std::min(nchunks, max_concurrent > 0 ? max_concurrent : 64));
    StreamGuard guard;
    if ((rc = guard.acquire())) return rc;
    const hipStream_t stream = guard.s;
    Events<3> ev;
    if ((rc = ev.create())) return rc;
    struct Prep {
        std::vector<int32_t> h_st, h_pos, h_sc;
        std::vector<int64_t> h_slots;
        int64_t N = 0; int Su = 0; int64_t ldc = 0;
        DevBuf cost;                                  // one row per spot that receives cells (Su x ldc)
        std::vector<int32_t> colsol, rowpos_u;        // rowpos_u[i] = stored row (= index into h_st) of LAP row i
        std::vector<int64_t> ones;
        bool identity = true;
        float ms_gather = 0, ms_gemm = 0;
    };
    int first = CYTO_OK;
    for (int lo = 0; lo < nchunks; lo += conc) {
        const int cnt = std::min(conc, nchunks - lo);
        std::vector<Prep> pp((size_t)cnt);
        DevBuf zst, zsc, dsc, dst;                    // gathered operands, reused by the chunks of this round
        size_t cap_st = 0, cap_sc = 0, cap_isc = 0, cap_ist = 0;
        auto build_one = [&](int k) -> int {                  // the cost matrix of chunk lo + k; a status other than OK ends the call
            int rc = CYTO_OK;
            cyto_chunk &ch = chunks[lo + k];
            Prep &p = pp[(size_t)k];
            ch.status = CYTO_OK; ch.total_cost = 0.0;
            memset(&ch.info, 0, sizeof ch.info);
            const int nst = ch.idx_st ? ch.n_st : ctx->S;
            if (!ch.idx_sc || ch.n_sc <= 0 || !ch.slots || !ch.mapped_spot || nst <= 0) { ch.status = CYTO_ERR_BAD_ARG; return CYTO_OK; }
            for (int t = 0; t < nst && ch.status == CYTO_OK; t++) {
                if (ch.slots[t] < 0) { ch.status = CYTO_ERR_BAD_ARG; break; }
                const int64_t s_ = ch.idx_st ? ch.idx_st[t] : t;
                if (s_ < 0 || s_ >= ctx->S) { ch.status = CYTO_ERR_BAD_ARG; break; }
                if (ch.slots[t] > 0) { p.h_st.push_back((int32_t)s_); p.h_pos.push_back(t); p.h_slots.push_back(ch.slots[t]); p.N += ch.slots[t]; }
            }
            if (ch.status == CYTO_OK && p.N != ch.n_sc) ch.status = CYTO_ERR_BAD_ARG;          // the LAP must be square
            p.h_sc.resize((size_t)std::max(ch.n_sc, 0));
            for (int c = 0; c < ch.n_sc && ch.status == CYTO_OK; c++) {
                if (ch.idx_sc[c] < 0 || ch.idx_sc[c] >= ctx->C) { ch.status = CYTO_ERR_BAD_ARG; break; }
                p.h_sc[(size_t)c] = (int32_t)ch.idx_sc[c];
            }
            if (ch.status != CYTO_OK) return CYTO_OK;
            p.Su = (int)p.h_st.size();
            const int n_sc = ch.n_sc;
            const int64_t ldzst = round_up(p.Su, BM), ldzsc = round_up(n_sc, BN);
            p.ldc = round_up(n_sc, 4);
            const size_t need_st = (size_t)ctx->Gpad * ldzst * 4, need_sc = (size_t)ctx->Gpad * ldzsc * 4;
            // (a larger operand than any before: the stream is drained first, the old block goes back to the cache)
            if (need_st > cap_st) { CYTO_HIP(hipStreamSynchronize(stream)); if ((rc = zst.alloc(need_st, stream))) return rc; cap_st = need_st; }
            if (need_sc > cap_sc) { CYTO_HIP(hipStreamSynchronize(stream)); if ((rc = zsc.alloc(need_sc, stream))) return rc; cap_sc = need_sc; }
            if ((size_t)n_sc * 4 > cap_isc) { CYTO_HIP(hipStreamSynchronize(stream)); if ((rc = dsc.alloc((size_t)n_sc * 4, stream))) return rc; cap_isc = (size_t)n_sc * 4; }
            if ((size_t)p.Su * 4 > cap_ist) { CYTO_HIP(hipStreamSynchronize(stream)); if ((rc = dst.alloc((size_t)p.Su * 4, stream))) return rc; cap_ist = (size_t)p.Su * 4; }
            if ((rc = p.cost.alloc((size_t)p.Su * p.ldc * 4, stream))) return rc;
            stamp("index lists validated, buffers allocated");
            p.ones.assign((size_t)p.Su, 1);
            p.rowpos_u.resize((size_t)p.N);
            {
                int64_t r = 0;
                for (int t = 0; t < p.Su; t++) {
                    p.identity = p.identity && p.h_slots[(size_t)t] == 1;
                    for (int64_t e = 0; e < p.h_slots[(size_t)t]; e++) p.rowpos_u[(size_t)r++] = t;
                }
            }
            CYTO_HIP(hipEventRecord(ev[0], stream));
            CYTO_HIP(hipMemcpyAsync(dsc.p, p.h_sc.data(), (size_t)n_sc * 4, hipMemcpyHostToDevice, stream));
            CYTO_HIP(hipMemcpyAsync(dst.p, p.h_st.data(), (size_t)p.Su * 4, hipMemcpyHostToDevice, stream));
            CYTO_HIP(hipMemsetAsync(zst.p, 0, need_st, stream));
            CYTO_HIP(hipMemsetAsync(zsc.p, 0, need_sc, stream));
            const int nblk = (ctx->Gpad + GB - 1) / GB;
            hipLaunchKernelGGL(gather_columns, dim3((p.Su + 255) / 256, nblk), dim3(256), 0, stream, ctx->Gpad, p.Su, ctx->zst.as<float>(),
                               ctx->ldst, dst.as<int32_t>(), zst.as<float>(), ldzst);
            hipLaunchKernelGGL(gather_columns, dim3((n_sc + 255) / 256, nblk), dim3(256), 0, stream, ctx->Gpad, n_sc, ctx->zsc.as<float>(),
                               ctx->ldsc, dsc.as<int32_t>(), zsc.as<float>(), ldzsc);
            CYTO_HIP(hipGetLastError());
            CYTO_HIP(hipEventRecord(ev[1], stream));
            double ms_gemm = 0;
            // (cost_gemm synchronises the stream: the host index vectors and the operands are free again afterwards)
            if ((rc = cyto_cost_metric(ctx->metric, ctx->Gpad, p.Su, n_sc, zst.as<float>(), ldzst, zsc.as<float>(), ldzsc, p.ones.data(),
                                       p.cost.as<float>(), p.ldc, &ms_gemm, ctx->device_id, stream)))
                return rc;
            (void)hipEventElapsedTime(&p.ms_gather, ev[0], ev[1]);
            p.ms_gemm = (float)ms_gemm;
            stamp("gathers + contraction done (stream synchronised)");
            p.colsol.resize((size_t)p.N);
            return CYTO_OK;
        };
        // the LAPs of chunks [k0, k1) of this round, together (runs on a worker thread while the next group's costs are built)
        auto solve_range = [&](int k0, int k1) {
        std::vector<int> ids;
        for (int k = k0; k < k1; k++) if (chunks[lo + k].status == CYTO_OK) ids.push_back(k);
        const int nl = (int)ids.size();
        if (nl) {
            std::vector<int> nn((size_t)nl), stat((size_t)nl, CYTO_OK), nus((size_t)nl, 0);
            std::vector<const int32_t *> rmaps((size_t)nl, nullptr);
            std::vector<const float *> cst((size_t)nl);
            std::vector<int64_t> lds_((size_t)nl);
            std::vector<int32_t *> cs((size_t)nl);
            std::vector<double> tot((size_t)nl);
            std::vector<cyto_lap_info> li((size_t)nl);
            for (int q = 0; q < nl; q++) {
                Prep &p = pp[(size_t)ids[(size_t)q]];
                nn[(size_t)q] = (int)p.N; cst[(size_t)q] = p.cost.as<float>(); lds_[(size_t)q] = p.ldc; cs[(size_t)q] = p.colsol.data();
                if (!p.identity) { rmaps[(size_t)q] = p.rowpos_u.data(); nus[(size_t)q] = p.Su; }
            }
            stamp("LAP batch begins");
            (void)lap_batch_any(nl, nn.data(), cst.data(), lds_.data(), 1, rmaps.data(), nus.data(), nullptr, cs.data(), nullptr, nullptr,
                                tot.data(), li.data(), stat.data(), nl, ctx->device_id);
            stamp("LAP batch returned");
            for (int q = 0; q < nl; q++) {
                const int k = ids[(size_t)q];
                cyto_chunk &ch = chunks[lo + k];
                Prep &p = pp[(size_t)k];
                ch.status = stat[(size_t)q];
                if (ch.status != CYTO_OK) continue;
                // location_repeat[y] (cytospace.py:331): LAP row -> position in the chunk's spot list
                for (int c = 0; c < ch.n_sc; c++) ch.mapped_spot[c] = p.h_pos[(size_t)p.rowpos_u[(size_t)p.colsol[(size_t)c]]];
                ch.total_cost = tot[(size_t)q];
                ch.info.ms_standardize = p.ms_gather;     // here: the two column gathers (the transforms ran once, at context creation)
                ch.info.ms_gemm = p.ms_gemm;
                ch.info.lap = li[(size_t)q];
                ch.info.gemm_flops = 2.0 * ctx->Gpad * (double)p.Su * (double)ch.n_sc;
            }
        }
        };
        // Groups of chunks: while the LAPs of a group run (latency-bound rounds and searches that leave most of the chip idle), the
        // main thread builds the next group's costs.  A group's LAPs go through lap_batch_any as one batch (its own streams and host
        // threads).  Measured (gpurun_out/r04ad against profiles/r04ac_bench.json): 50 single-cell-mode chunks with their 500-gene cost
        // builds 0.18 -> 0.15 s; 64 c4 chunks with 5 000-gene builds 0.75 s either way -- the contractions take 540 instead of 451 ms
        // with LAP rounds among them, and a round whose launches queue behind contraction workgroups takes as much longer.
        const int per_group = cnt >= 16 ? std::max(8, (cnt + 3) / 4) : cnt;
        auto solve_guarded = [&](int k0, int k1) {
            try { solve_range(k0, k1); }
            catch (...) { for (int k = k0; k < k1; k++) if (chunks[lo + k].status == CYTO_OK) chunks[lo + k].status = CYTO_ERR_NOMEM; }
        };
        {
            std::vector<std::thread> workers;
            struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (std::thread &x : t) if (x.joinable()) x.join(); } } joiner{workers};
            int g0 = 0;
            for (int k = 0; k < cnt; k++) {
                if ((rc = build_one(k))) return rc;             // (the joiner waits for the groups already under way)
                if (k + 1 - g0 >= per_group || k + 1 == cnt) {
                    const int a0 = g0, a1 = k + 1;
                    g0 = a1;
                    if (a1 == cnt) { solve_guarded(a0, a1); break; }      // the last group: on this thread
                    bool started = false;
                    try { workers.emplace_back(solve_guarded, a0, a1); started = true; } catch (...) { started = false; }
                    if (!started) solve_guarded(a0, a1);
                }
            }
        }
        for (int k = 0; k < cnt; k++) if (!first && chunks[lo + k].status) first = chunks[lo + k].status;
        stamp("round of chunks done; buffers go back to the block cache");
    }
    stamp("return");
    return first;
}

int cyto_assign_pearson(int G, int C, int S, const double *sc, const double *st, const int64_t *slots, int already_normalized,
                        int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id) {
    return cyto_assign_metric(CYTO_METRIC_PEARSON, G, C, S, sc, st, slots, already_normalized, mapped_spot, total_cost, info, device_id);
}

int cyto_assign_metric(int metric, int G, int C, int S, const double *sc, const double *st, const int64_t *slots, int already_normalized,
                       int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id) {
    return cyto_assign_metric_typed(metric, G, C, S, sc, st, 1, slots, already_normalized, mapped_spot, total_cost, info, device_id);
}

int cyto_ctx_create(int metric, int G, int C, int S, const double *sc, const double *st, int already_normalized, int device_id,
                    cyto_expr_ctx **out) {
    return cyto_ctx_create_typed(metric, G, C, S, sc, st, 1, already_normalized, device_id, out);
}

}  // extern "C"
