// Point-landmark log-likelihood: value, gradient and exact Hessian, assembled
// into the dense state gradient / Hessian by one kernel launch.
//
// Replaces PointLogLikelihood::operator() (reference src/model.cpp:408-437),
// whose gradient/Hessian come from n and n(n+1)/2 forward-mode passes; here
// every landmark's 9x9 block is evaluated in closed form by one lane.
//
// Work decomposition (grid.x = 1 + panels, grid.y = batch instance):
//   block 0  "camera block": loops over all L landmarks, 64 per wavefront.
//            Each lane evaluates one landmark, stores its rows of the six dense
//            pose columns (rows 12+3j.. of columns 6..11, contiguous across
//            lanes), its gradient entries and cull code, and accumulates the
//            28 pose sums (6x6 block, 6-gradient, value) which are reduced by
//            a wavefront butterfly + a fixed-order LDS pass.  It also writes
//            the all-zero velocity columns 0..5.
//   block p  "landmark panel": P consecutive landmarks = 3P columns of H.
//            One lane per landmark re-evaluates that landmark with the same
//            instruction sequence as block 0 (so H comes out exactly
//            symmetric) and parks its 6x3 cross block and 3x3 diagonal block
//            in LDS; then the whole block streams the 3P columns to HBM in
//            row order, 512 contiguous bytes per wavefront store, selecting
//            the parked values on the 9 non-zero rows of each column and zero
//            elsewhere (H is >97 % zeros but is a dense output: the
//            reference's callee resizes and fills it).  No cache line is
//            written twice or partially, except at panel edges.
// The kernel is HBM-write bound: 8 n^2 bytes of H per instance.
#include "landmark_math.hpp"

#include <mutex>

namespace vslam {

struct LoglikArgs {
    const double *x; size_t sx;
    const double *y; size_t sy;
    const uint8_t *vis; size_t sv;
    double *value;
    double *g; size_t sg;
    double *H; size_t sH;
    int32_t *code;
    int ldH, n, L, P;
    double is2, lconst;
    const int *active;       // optional: batch instances whose entry is 0 are skipped
};

constexpr int kThreads = 256;

// Cooperative zero fill of p[0..count) by the whole block; p is 8-byte aligned.
__device__ __forceinline__ void block_zero(double *p, size_t count)
{
    if (count == 0) return;
    const int tid = threadIdx.x;
    size_t head = (reinterpret_cast<uintptr_t>(p) & 8) ? 1 : 0;
    if (head && tid == 0) p[0] = 0.0;
    double2 *p2 = reinterpret_cast<double2 *>(p + head);
    size_t n2 = (count - head) >> 1;
    const double2 z = make_double2(0.0, 0.0);
    size_t i = tid;
    for (; i + 3 * kThreads < n2; i += 4 * kThreads) {
        p2[i] = z; p2[i + kThreads] = z; p2[i + 2 * kThreads] = z; p2[i + 3 * kThreads] = z;
    }
    for (; i < n2; i += kThreads) p2[i] = z;
    if (((count - head) & 1) && tid == 0) p[count - 1] = 0.0;
}

__device__ __forceinline__ void block_zero_cols(double *H, int ldH, int n, int c0, int c1)
{
    if (c1 <= c0) return;
    if (ldH == n) block_zero(H + (size_t)c0 * ldH, (size_t)(c1 - c0) * n);
    else
        for (int c = c0; c < c1; ++c) block_zero(H + (size_t)c * ldH, n);
}

template <int ORDER>
__global__ __launch_bounds__(kThreads) void point_loglik_kernel(CamK cam, LoglikArgs a)
{
    extern __shared__ double vals[];                   // [P][3 columns][9 rows] parked non-zeros of a landmark panel
    const int b = blockIdx.y, tid = threadIdx.x;
    if (a.active && !a.active[b]) return;
    const double *x = a.x + (size_t)b * a.sx;
    const double *y = a.y + (size_t)b * a.sy;
    const uint8_t *vis = a.vis + (size_t)b * a.sv;
    double *g = (ORDER >= 1) ? a.g + (size_t)b * a.sg : nullptr;
    double *H = (ORDER >= 2) ? a.H + (size_t)b * a.sH : nullptr;
    const int n = a.n, L = a.L, ldH = a.ldH;
    const bool is_cam = blockIdx.x == 0;

    int jb, je;
    if (is_cam) { jb = 0; je = L; }
    else { jb = (blockIdx.x - 1) * a.P; je = min(L, jb + a.P); }

    if (ORDER >= 2 && is_cam) block_zero_cols(H, ldH, n, 0, 6);

    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.0;

    if (jb + (tid & ~63) < je) {                       // wave-uniform: whole wavefronts enter
        const Pose pose = make_pose_wave(x + 6);
        for (int j = jb + tid; j < je; j += kThreads) {
            Terms t;
            landmark_terms<ORDER>(cam, pose, x + 12 + 3 * j, y + 2 * j, vis[j] != 0, a.is2, a.lconst, t);
            const double Nf[3][3] = {{t.N[0], t.N[1], t.N[2]}, {t.N[1], t.N[3], t.N[4]}, {t.N[2], t.N[4], t.N[5]}};
            if (is_cam) {
                acc[0] += t.ell;
                if (a.code) a.code[(size_t)b * L + j] = t.code;
                if (ORDER >= 1) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { acc[1 + i] += t.gr[i]; acc[4 + i] += t.gT[i]; g[12 + 3 * j + i] = t.gr[i]; }
                }
                if (ORDER >= 2) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { acc[7 + i] += t.N[i]; acc[22 + i] += t.F[i]; }
#pragma unroll
                    for (int i = 0; i < 9; ++i) acc[13 + i] += t.E[i];
                    // rows 12+3j.. of the pose columns: H[r_i][c_a] = -N[i][a], H[r_i][Theta_k] = E[i][k]
                    // (3 consecutive doubles per lane, consecutive lanes adjacent: contiguous stores)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double *col = H + (size_t)(6 + c) * ldH + 12 + 3 * j;
                        double *colT = H + (size_t)(9 + c) * ldH + 12 + 3 * j;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { col[i] = -Nf[i][c]; colT[i] = t.E[i * 3 + c]; }
                    }
                }
            } else if (ORDER >= 2) {
                // column 12+3j+i: rows 6..8 = -N[:,i], rows 9..11 = E[i][:], rows 12+3j.. = N[:,i]
                double *v = vals + (size_t)(j - jb) * 27;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        v[i * 9 + r] = -Nf[r][i];
                        v[i * 9 + 3 + r] = t.E[i * 3 + r];
                        v[i * 9 + 6 + r] = Nf[r][i];
                    }
            }
        }
    }
    if (ORDER >= 2 && !is_cam) {
        __syncthreads();
        // stream the panel: flat index over (column, row), one double per lane per store
        const int ncol = 3 * (je - jb), c0 = 12 + 3 * jb;
        int col = 0, row = tid;
        while (row >= n) { row -= n; ++col; }
        while (col < ncol) {
            const int jl = col / 3, d0 = 12 + 3 * (jb + jl);
            const double *v = vals + (size_t)col * 9;           // == jl*27 + (col%3)*9
            double out = 0.0;
            if (row >= 6 && row < 12) out = v[row - 6];
            else if (row >= d0 && row < d0 + 3) out = v[6 + row - d0];
            H[(size_t)(c0 + col) * ldH + row] = out;
            row += kThreads;
            while (row >= n) { row -= n; ++col; }
        }
    }
    if (!is_cam) return;

    // ---- reduce the 28 pose sums: wavefront butterfly, then 4 partials in fixed order ----
    __shared__ double red[kThreads / 64][28];
    constexpr int NRED = (ORDER == 0) ? 1 : (ORDER == 1) ? 7 : 28;
#pragma unroll
    for (int i = 0; i < NRED; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NRED; ++i) red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    __shared__ double tot[28];
    if (tid < NRED) tot[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    __syncthreads();
    if (tid == 0) a.value[b] = tot[0];
    if (ORDER >= 1 && tid < 12) {
        double v = 0.0;
        if (tid >= 6 && tid < 9) v = -tot[1 + (tid - 6)];
        else if (tid >= 9) v = tot[4 + (tid - 9)];
        g[tid] = v;
    }
    if (ORDER >= 2 && tid < 72) {
        // head rows 0..11 of columns 6..11
        const int c = tid / 12, r = tid % 12;          // column 6+c, row r
        const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        double v = 0.0;
        if (r >= 6) {
            const int rr = r - 6;
            if (c < 3 && rr < 3) v = tot[7 + sym[rr][c]];              // sum N
            else if (c >= 3 && rr >= 3) v = tot[22 + sym[rr - 3][c - 3]];  // sum F
            else if (c < 3) v = -tot[13 + c * 3 + (rr - 3)];           // H[Theta_k][c_i] = -E[i][k]
            else v = -tot[13 + rr * 3 + (c - 3)];                      // H[c_i][Theta_k] = -E[i][k]
        }
        H[(size_t)(6 + c) * ldH + r] = v;
    }
}

// ---------------------------------------------------------------------------
// Two-launch form for large outputs.  In the fused kernel every workgroup first
// evaluates its landmarks (a ~5 us dependent fp64 chain at 250 VGPRs, two
// workgroups per CU) and only then streams, and co-resident workgroups do so in
// phase, so the store pipes idle during every evaluation.  Here a small kernel
// evaluates all landmarks once (N and E blocks to a structure-of-arrays record,
// pose sums reduced per block), and a low-register streaming kernel at full
// occupancy assembles H from the records.
// ---------------------------------------------------------------------------
constexpr int kRecFields = 15;        // N[6] then E[9]

struct TermsArgs {
    LoglikArgs a;
    double *rec;       // [batch][kRecFields][L]
    double *part;      // [batch][nblk][28]
    int nblk;
    int slice_bytes;   // bytes of H per workgroup of the assemble kernel (a multiple of 128)
};

// Sums each of v[0..31] over the 64 lanes of the wavefront in 32 exchanges instead of 6 per value: at every step a lane
// keeps half of its values and hands the other half to the partner that keeps those (offsets 32, 16, 8, 4, 2), so the
// number of values per lane halves as the number of lanes per sum doubles; a last exchange joins the two lanes of a pair.
// Lane l returns the total of v[l >> 1].  Fixed order of additions.
VSLAM_DEV double wave_sum32(const double (&v)[32], int lane)
{
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4, h1 = lane & 2;
    double a[16], b[8], c[4], d[2];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = (h5 ? v[m + 16] : v[m]) + __shfl_xor(h5 ? v[m] : v[m + 16], 32, 64);
#pragma unroll
    for (int m = 0; m < 8; ++m) b[m] = (h4 ? a[m + 8] : a[m]) + __shfl_xor(h4 ? a[m] : a[m + 8], 16, 64);
#pragma unroll
    for (int m = 0; m < 4; ++m) c[m] = (h3 ? b[m + 4] : b[m]) + __shfl_xor(h3 ? b[m] : b[m + 4], 8, 64);
#pragma unroll
    for (int m = 0; m < 2; ++m) d[m] = (h2 ? c[m + 2] : c[m]) + __shfl_xor(h2 ? c[m] : c[m + 2], 4, 64);
    double e = (h1 ? d[1] : d[0]) + __shfl_xor(h1 ? d[0] : d[1], 2, 64);
    e += __shfl_xor(e, 1, 64);
    return e;
}

// One wavefront per workgroup: 64 landmarks of one instance.  Nothing is shared between wavefronts (no barrier, no
// LDS), a launch of B instances of L landmarks is B * ceil(L / 64) independent waves spread over the whole chip.
constexpr int kTermThreads = 64;
__global__ __launch_bounds__(kTermThreads) void loglik_terms_kernel(CamK cam, TermsArgs t)
{
    const LoglikArgs &a = t.a;
    const int b = blockIdx.y, lane = threadIdx.x, L = a.L;
    if (a.active && !a.active[b]) return;
    const double *x = a.x + (size_t)b * a.sx;
    const double *y = a.y + (size_t)b * a.sy;
    const uint8_t *vis = a.vis + (size_t)b * a.sv;
    double *g = a.g + (size_t)b * a.sg;
    double *rec = t.rec + (size_t)b * kRecFields * L;
    const int j = blockIdx.x * kTermThreads + lane;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    // the landmark, its measurement and its flag are requested before the pose's sines and cosines are worked out: one
    // trip to memory on the wave's critical path instead of two
    const int jc = min(j, L - 1);
    const double rl[3] = {x[12 + 3 * jc], x[12 + 3 * jc + 1], x[12 + 3 * jc + 2]}, yl[2] = {y[2 * jc], y[2 * jc + 1]};
    const bool seen = vis[jc] != 0;
    const Pose pose = make_pose_wave(x + 6);
    if (j < L) {
        Terms tm;
        landmark_terms<2>(cam, pose, rl, yl, seen, a.is2, a.lconst, tm);
        acc[0] = tm.ell;
#pragma unroll
        for (int i = 0; i < 3; ++i) { acc[1 + i] = tm.gr[i]; acc[4 + i] = tm.gT[i]; g[12 + 3 * j + i] = tm.gr[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) { acc[7 + i] = tm.N[i]; acc[22 + i] = tm.F[i]; rec[(size_t)i * L + j] = tm.N[i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) { acc[13 + i] = tm.E[i]; rec[(size_t)(6 + i) * L + j] = tm.E[i]; }
        if (a.code) a.code[(size_t)b * L + j] = tm.code;
    }
    const double tot = wave_sum32(acc, lane);
    if (!(lane & 1) && (lane >> 1) < 28) t.part[((size_t)b * t.nblk + blockIdx.x) * 28 + (lane >> 1)] = tot;
}

// value of H[row][column 12+3j+i] on its non-zero rows, from the records
__device__ __forceinline__ double rec_N(const double *rec, int L, int j, int r, int c)
{
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int f = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);      // xx xy xz yy yz zz
    return rec[(size_t)f * L + j];
}

// Streaming assembly of H from the records.  H (n x n, ld == n) is treated as a flat
// array; each workgroup owns one 32 KB slice whose boundaries fall on 128-byte lines of
// the absolute address, and writes it with 16-byte stores in address order, so every
// cache line is written whole by one wavefront store (partial-line writes cost a
// read-modify-write at the memory side) and the workgroups in flight cover one compact
// window of HBM.  An element is zero unless it lies in a pose column (6..11) or on one
// of the 9 non-zero rows of a landmark column; only those take the lookup path.
constexpr int kSliceBytes = 32768;

// srec: LDS copy of the records of landmarks j_lo.. (field-major, stride kRecFields per landmark)
__device__ __forceinline__ double srec_N(const double *srec, int jl, int r, int c)
{
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    return srec[jl * kRecFields + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5))];
}

__device__ __forceinline__ double assemble_value(const TermsArgs &t, const double *rec, const double *part_b,
                                                 const double *srec, int j_lo, int col, int row)
{
    const int L = t.a.L;
    if (col < 6 || row < 6) return 0.0;
    if (col < 12) {
        const int c = col - 6;
        if (row >= 12) {                               // H[r_i][pose c]
            const int j = (row - 12) / 3, i = (row - 12) - 3 * j;
            return c < 3 ? -rec_N(rec, L, j, i, c) : rec[(size_t)(6 + i * 3 + (c - 3)) * L + j];
        }
        const int rr = row - 6;                        // 6x6 pose block: sums over all landmarks
        int f; double sgn = 1.0;
        if (c < 3 && rr < 3) { const int lo = min(rr, c), hi = max(rr, c); f = 7 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)); }
        else if (c >= 3 && rr >= 3) { const int lo = min(rr, c) - 3, hi = max(rr, c) - 3; f = 22 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)); }
        else if (c < 3) { f = 13 + c * 3 + (rr - 3); sgn = -1.0; }
        else { f = 13 + rr * 3 + (c - 3); sgn = -1.0; }
        double s = 0.0;
        for (int k = 0; k < t.nblk; ++k) s += part_b[(size_t)k * 28 + f];
        return sgn * s;
    }
    const int j = (col - 12) / 3, i = (col - 12) - 3 * j, d0 = 12 + 3 * j, jl = j - j_lo;
    if (row < 9) return -srec_N(srec, jl, row - 6, i);                          // H[c_r][r_i]
    if (row < 12) return srec[jl * kRecFields + 6 + i * 3 + (row - 9)];         // H[Theta_r][r_i] = E[i][r]
    if (row >= d0 && row < d0 + 3) return srec_N(srec, jl, row - d0, i);
    return 0.0;
}

__device__ __forceinline__ bool assemble_special(int col, int row, int n)
{
    if (col < 6 || row < 6) return false;
    if (col < 12 || row < 12) return true;
    const int d0 = 12 + 3 * ((col - 12) / 3);
    return row >= d0 && row < d0 + 3;
}

constexpr int kAsmThreads = 256;        // 512 and 1 024 threads with slices of 32-128 KB measured 172-233 us against 155-160
__global__ __launch_bounds__(kAsmThreads) void loglik_assemble_kernel(TermsArgs t)
{
    extern __shared__ double srec[];                   // records of the landmarks whose columns this slice touches
    const LoglikArgs &a = t.a;
    const int b = blockIdx.y, tid = threadIdx.x, n = a.n, L = a.L;
    if (a.active && !a.active[b]) return;
    double *H = a.H + (size_t)b * a.sH;
    const double *rec = t.rec + (size_t)b * kRecFields * L;
    const double *part_b = t.part + (size_t)b * t.nblk * 28;
    if (blockIdx.x == 0 && tid < 13) {
        // value and the pose part of the gradient (landmark entries were written by the terms kernel)
        double *g = a.g + (size_t)b * a.sg;
        const int f = tid == 12 ? 0 : (tid < 6 ? 0 : (tid < 9 ? 1 + (tid - 6) : 4 + (tid - 9)));
        double s = 0.0;
        for (int k = 0; k < t.nblk; ++k) s += part_b[(size_t)k * 28 + f];
        if (tid == 12) a.value[b] = s;
        else g[tid] = tid < 6 ? 0.0 : (tid < 9 ? -s : s);
    }
    // this block's slice of the flat element range [0, n*n), cut at absolute 128-byte lines
    const uint32_t total = (uint32_t)n * (uint32_t)n;              // element indices fit 32 bits (the launcher checks n < 65536)
    const uintptr_t base = reinterpret_cast<uintptr_t>(H);
    const uintptr_t line0 = base & ~uintptr_t(127);
    const uintptr_t lo = line0 + (uintptr_t)blockIdx.x * (uintptr_t)t.slice_bytes, hi = lo + (uintptr_t)t.slice_bytes;
    uint32_t e0 = lo <= base ? 0u : (uint32_t)((lo - base) >> 3);
    uint32_t e1 = hi <= base ? 0u : (uint32_t)((hi - base) >> 3);
    if (e1 > total) e1 = total;
    if (e0 >= e1) return;
    const int cA = (int)(e0 / n), cB = (int)((e1 - 1) / n);
    if (cA >= 12) {
        // The slice holds landmark columns only (all but the first two slices of an instance).  A column 12 + 3 j + i
        // has nine non-zero rows: 6..11 (the pose rows: -N[.][i] and E[i][.]) and the three rows d0 = 3 (col / 3) .. d0 + 2
        // of its own landmark (N[.][i]).  They are staged per column, in row order, so that an element finds its value
        // with two comparisons and one LDS read and no wavefront takes a slow path for the sake of a few lanes.
        double *scol = srec;                                                    // [column of the slice][9]
        for (int q = tid; q < (cB - cA + 1) * 9; q += kAsmThreads) {
            const int cl = q / 9, f = q - 9 * cl, col = cA + cl, j = (col - 12) / 3, i = col - 12 - 3 * j;
            scol[q] = f < 3 ? -rec_N(rec, L, j, f, i) : (f < 6 ? rec[(size_t)(6 + i * 3 + (f - 3)) * L + j] : rec_N(rec, L, j, f - 6, i));
        }
        __syncthreads();
        auto in_column = [&](const double *cv, int d0, int row) -> double {
            const int t = row - d0;
            const bool ok = (unsigned)(row - 6) < 6u || (unsigned)t < 3u;
            return ok ? cv[row < 12 ? row - 6 : t + 6] : 0.0;
        };
        auto value = [&](int col, int row) -> double { return in_column(scol + (col - cA) * 9, 3 * (col / 3), row); };
        uint32_t e = e0;
        if ((base + ((uintptr_t)e << 3)) & 8) {                   // head single (only possible at the instance start, which lies in a pose column)
            if (tid == 0) { const int col = (int)(e / n), row = (int)(e - (uint32_t)col * n); H[e] = value(col, row); }
            ++e;
        }
        const uint32_t npair = (e1 - e) >> 1;
        const uint32_t idx = e + 2u * (uint32_t)tid;
        int col = (int)(idx / n), row = (int)(idx - (uint32_t)col * n);
        for (uint32_t k = tid; k < npair; k += kAsmThreads) {
            const double *cv = scol + (col - cA) * 9;
            const int d0 = 3 * (int)(((unsigned)col * 43691u) >> 17);           // 3 * (col / 3) for col < 65536
            double2 o;
            o.x = in_column(cv, d0, row);
            o.y = row + 1 == n ? 0.0 : in_column(cv, d0, row + 1);              // a pair that ends in row 0 of the next column: zero there
            *reinterpret_cast<double2 *>(H + e + 2 * k) = o;
            row += 2 * kAsmThreads;
            while (row >= n) { row -= n; ++col; }
        }
        if (((e1 - e) & 1) && tid == 0) {              // tail single (only at the instance end)
            const uint32_t el = e1 - 1;
            const int c = (int)(el / n), r = (int)(el - (uint32_t)c * n);
            H[el] = value(c, r);
        }
        return;
    }
    int j_lo = 0;
    {
        if (cB >= 12) {
            j_lo = cA >= 12 ? (cA - 12) / 3 : 0;
            const int j_hi = (cB - 12) / 3;
            for (int q = tid; q < (j_hi - j_lo + 1) * kRecFields; q += kAsmThreads) {
                const int jl = q / kRecFields, f = q - jl * kRecFields;
                srec[q] = rec[(size_t)f * L + j_lo + jl];
            }
        }
        __syncthreads();
    }
    // head single if the slice starts on an odd 8-byte boundary (only possible at the instance start)
    uint32_t e = e0;
    if ((base + ((uintptr_t)e << 3)) & 8) {
        if (tid == 0) { const int col = (int)(e / n), row = (int)(e - (uint32_t)col * n); H[e] = assemble_value(t, rec, part_b, srec, j_lo, col, row); }
        ++e;
    }
    const uint32_t npair = (e1 - e) >> 1;
    const uint32_t idx = e + 2u * (uint32_t)tid;
    int col = (int)(idx / n), row = (int)(idx - (uint32_t)col * n);
    for (uint32_t k = tid; k < npair; k += kAsmThreads) {
        int col2 = col, row2 = row + 1;
        if (row2 == n) { row2 = 0; ++col2; }
        double2 o = make_double2(0.0, 0.0);
        if (assemble_special(col, row, n) || assemble_special(col2, row2, n)) {
            o.x = assemble_value(t, rec, part_b, srec, j_lo, col, row);
            o.y = assemble_value(t, rec, part_b, srec, j_lo, col2, row2);
        }
        *reinterpret_cast<double2 *>(H + e + 2 * k) = o;
        row += 2 * kAsmThreads;
        while (row >= n) { row -= n; ++col; }
    }
    if (((e1 - e) & 1) && tid == 0) {                  // tail single (only at the instance end)
        const uint32_t el = e1 - 1;
        const int c = (int)(el / n), r = (int)(el - (uint32_t)c * n);
        H[el] = assemble_value(t, rec, part_b, srec, j_lo, c, r);
    }
}

static int pick_panel(int L, int batch)
{
    // aim for a few thousand workgroups so all 256 CUs stream stores
    long total = (long)L * batch;
    int P = (int)((total + 4095) / 4096);
    if (P < 1) P = 1;
    if (P > kThreads) P = kThreads;
    return P;
}

// Scratch of the entry point that allocates nothing: the records and per-block pose sums of the
// two-launch form for at most kStaticLandmarks landmarks over all instances of a call and
// kStaticBlocks256 (instance, 256-landmark block) pairs (the sums are kept per 64 landmarks: four times as many
// slots).  One copy per GPU, held until the process ends.
constexpr size_t kStaticLandmarks = 81920, kStaticBlocks256 = 4096, kStaticBlocks = 4 * kStaticBlocks256;    // blocks of 64 landmarks
constexpr size_t kStaticRecDoubles = kStaticLandmarks * kRecFields, kStaticPartDoubles = kStaticBlocks * 28;
static __device__ __align__(256) double scratch[kStaticRecDoubles + kStaticPartDoubles];
// Device address of `scratch` per GPU (the array's name is a host-side stand-in, not a device address).
constexpr int kMaxDevices = 64;
static double *g_static_scratch[kMaxDevices];
static std::mutex g_static_scratch_mutex;

vslam_status loglik_static_scratch_init(vslam_ctx *ctx)
{
    if (ctx->device < 0 || ctx->device >= kMaxDevices) return VSLAM_OK;      // beyond the table: the no-alloc entry reports it
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (g_static_scratch[ctx->device]) return VSLAM_OK;
    double *ptr = nullptr;
    VSLAM_HIP_TRY(ctx, hipGetSymbolAddress(reinterpret_cast<void **>(&ptr), HIP_SYMBOL(scratch)));
    if (!ptr) return fail(ctx, VSLAM_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the static scratch array");
    g_static_scratch[ctx->device] = ptr;
    return VSLAM_OK;
}

static vslam_status launch_point_loglik_impl(vslam_ctx *ctx, const CamK &cam, int batch, int n,
                                             const double *x, size_t sx, const double *y, size_t sy,
                                             const uint8_t *vis, size_t sv, double sigma,
                                             double *value, double *g, size_t sg, double *H, int ldH, size_t sH,
                                             int32_t *code, const int *active, double *static_scratch)
{
    // n = 12 is the map without a landmark yet (the reference evaluates the likelihood on every frame, src/SLAM.cpp:489-495,
    // and its loop over param.landmarks, src/model.cpp:392, is then empty): value 0, zero gradient and Hessian; y and
    // visible have no entries and may be null
    if (n < 12 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: n=%d is not 12+3L", n);
    if (!x || !value || (n > 12 && (!y || !vis))) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: null input/value pointer");
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: batch=%d out of range", batch);
    if (H && !g) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: H requested without g");
    if (H && ldH < n) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: ldH=%d < n=%d", ldH, n);
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: sigma must be positive");
    LoglikArgs a;
    a.x = x; a.sx = sx; a.y = y; a.sy = sy; a.vis = vis; a.sv = sv;
    a.value = value; a.g = g; a.sg = sg; a.H = H; a.sH = sH; a.code = code; a.active = active;
    a.ldH = ldH; a.n = n; a.L = (n - 12) / 3;
    a.P = pick_panel(a.L, batch);
    a.is2 = 1.0 / (sigma * sigma);
    a.lconst = -std::log(2.0 * 3.14159265358979323846264338327950288) - 2.0 * std::log(sigma);
    const int panels = (a.L + a.P - 1) / a.P;
    dim3 block(kThreads);
    // With the static scratch every Hessian goes through the two-launch form (the one that needs scratch).
    const bool two_launch = H && ldH == n && a.L > 0 && n < 65536 && (static_scratch || (size_t)batch * n * n * 8 >= ((size_t)32 << 20));
    if (two_launch) {
        int slice_bytes = kSliceBytes;
        if (const char *e = getenv("VSLAM_LOGLIK_SLICE")) { const int v = atoi(e); if (v >= 1024 && v <= (1 << 20)) slice_bytes = v & ~127; }     // A/B runs
        const int slices = (int)(((size_t)n * n * 8 + 127 + slice_bytes - 1) / slice_bytes) + 1;
        TermsArgs t;
        t.a = a;
        t.slice_bytes = slice_bytes;
        t.nblk = (a.L + kTermThreads - 1) / kTermThreads;
        const size_t rec_doubles = (size_t)batch * kRecFields * a.L, part_doubles = (size_t)batch * t.nblk * 28;
        if (static_scratch) {
            if ((size_t)batch * a.L > kStaticLandmarks || (size_t)batch * ((a.L + 255) / 256) > kStaticBlocks256)
                return fail(ctx, VSLAM_ERR_CAPACITY,
                            "point_loglik (no allocation): %d instances x %d landmarks exceed the static scratch (%zu landmarks, %zu blocks of 256)",
                            batch, a.L, kStaticLandmarks, kStaticBlocks256);
            t.rec = static_scratch;                      // every word the kernels read is written first: nothing to clear
            t.part = static_scratch + kStaticRecDoubles;
        } else {
            VSLAM_TRY(scratch_reserve(ctx, Arena::pad(rec_doubles * 8) + Arena::pad(part_doubles * 8)));
            Arena ar(ctx->scratch);
            t.rec = ar.take<double>(rec_doubles);
            t.part = ar.take<double>(part_doubles);
        }
        const bool timed = ctx->time_kernels && ctx->tev[0];
        if (timed) VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->tev[0], ctx->stream));
        hipLaunchKernelGGL(loglik_terms_kernel, dim3(t.nblk, batch), dim3(kTermThreads), 0, ctx->stream, cam, t);
        if (timed) VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->tev[1], ctx->stream));
        // records of the landmarks of a slice (the slices that reach into the pose columns) or nine values per column of it
        const size_t lds_rec = (size_t)((slice_bytes / 8) / (3 * n) + 3) * kRecFields, lds_col = (size_t)((slice_bytes / 8) / n + 2) * 9;
        const size_t lds = (lds_rec > lds_col ? lds_rec : lds_col) * sizeof(double);

        hipLaunchKernelGGL(loglik_assemble_kernel, dim3(slices, batch), dim3(kAsmThreads), lds, ctx->stream, t);
        if (timed) { VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->tev[2], ctx->stream)); ctx->tev_recorded = true; }
    } else if (H) {
        dim3 grid(1 + panels, batch);
        hipLaunchKernelGGL(point_loglik_kernel<2>, grid, block, (size_t)a.P * 27 * sizeof(double), ctx->stream, cam, a);
    } else if (g) {
        hipLaunchKernelGGL(point_loglik_kernel<1>, dim3(1, batch), block, 0, ctx->stream, cam, a);
    } else {
        hipLaunchKernelGGL(point_loglik_kernel<0>, dim3(1, batch), block, 0, ctx->stream, cam, a);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

vslam_status launch_point_loglik(vslam_ctx *ctx, const CamK &cam, int batch, int n,
                                 const double *x, size_t sx, const double *y, size_t sy,
                                 const uint8_t *vis, size_t sv, double sigma,
                                 double *value, double *g, size_t sg, double *H, int ldH, size_t sH,
                                 int32_t *code, const int *active)
{
    return launch_point_loglik_impl(ctx, cam, batch, n, x, sx, y, sy, vis, sv, sigma, value, g, sg, H, ldH, sH, code, active, nullptr);
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_ctx_set_kernel_timing(vslam_ctx *ctx, int on)
{
    if (!ctx) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on)
        for (int i = 0; i < 3; ++i)
            if (!ctx->tev[i]) VSLAM_HIP_TRY(ctx, hipEventCreate(&ctx->tev[i]));
    ctx->time_kernels = on != 0;
    ctx->tev_recorded = false;
    return VSLAM_OK;
}

extern "C" vslam_status vslam_ctx_last_point_loglik_times(vslam_ctx *ctx, double *terms_ms, double *assemble_ms)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!terms_ms || !assemble_ms) return fail(ctx, VSLAM_ERR_ARG, "last_point_loglik_times: null pointer");
    if (!ctx->tev_recorded) return fail(ctx, VSLAM_ERR_ARG, "last_point_loglik_times: no timed two-kernel call since timing was switched on");
    VSLAM_HIP_TRY(ctx, hipEventSynchronize(ctx->tev[2]));
    float a = 0.f, b = 0.f;
    VSLAM_HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->tev[0], ctx->tev[1]));
    VSLAM_HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->tev[1], ctx->tev[2]));
    *terms_ms = a; *assemble_ms = b;
    return VSLAM_OK;
}

extern "C" vslam_status vslam_point_loglik_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, int batch, int n,
                                                     const double *x, size_t stride_x, const double *y, size_t stride_y,
                                                     const uint8_t *visible, size_t stride_vis, double sigma,
                                                     double *value, double *g, size_t stride_g,
                                                     double *H, int ldH, size_t stride_H, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_point_loglik(ctx, k, batch, n, x, stride_x, y, stride_y, visible, stride_vis, sigma, value, g,
                               stride_g, H, ldH, stride_H, code, nullptr);
}

extern "C" vslam_status vslam_point_loglik_batch_dev_noalloc(vslam_ctx *ctx, const vslam_camera *cam, int batch, int n,
                                                             const double *x, size_t stride_x, const double *y, size_t stride_y,
                                                             const uint8_t *visible, size_t stride_vis, double sigma,
                                                             double *value, double *g, size_t stride_g,
                                                             double *H, int ldH, size_t stride_H, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    if (H && ldH != n) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik (no allocation): ldH=%d must equal n=%d", ldH, n);
    if (ctx->device < 0 || ctx->device >= kMaxDevices || !g_static_scratch[ctx->device])
        return fail(ctx, VSLAM_ERR_DEVICE, "point_loglik (no allocation): no static scratch address for device %d", ctx->device);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_point_loglik_impl(ctx, k, batch, n, x, stride_x, y, stride_y, visible, stride_vis, sigma, value, g,
                                    stride_g, H, ldH, stride_H, code, nullptr, g_static_scratch[ctx->device]);
}

extern "C" vslam_status vslam_point_loglik(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                           const double *y, const uint8_t *visible, double sigma,
                                           double *value, double *g, double *H, int ldH, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (n < 12 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: n=%d is not 12+3L", n);
    if (!x || !value || (n > 12 && (!y || !visible))) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: null pointer");
    if (H && ldH < n) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: ldH=%d < n=%d", ldH, n);
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int L = (n - 12) / 3;
    const size_t nn = (size_t)n * n;
    size_t bytes = Arena::pad(n * 8) * 2 + Arena::pad(2 * L * 8) + Arena::pad(L) + Arena::pad(L * 4) + Arena::pad(8) +
                   (H ? Arena::pad(nn * 8) : 0);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dg = ar.take<double>(n), *dy = ar.take<double>(2 * L);
    uint8_t *dv = ar.take<uint8_t>(L);
    int32_t *dc = ar.take<int32_t>(L);
    double *dval = ar.take<double>(1);
    double *dH = H ? ar.take<double>(nn) : nullptr;
    VSLAM_TRY(h2d(ctx, dx, x, n * 8));
    VSLAM_TRY(h2d(ctx, dy, y, 2 * L * 8));
    VSLAM_TRY(h2d(ctx, dv, visible, L));
    VSLAM_TRY(launch_point_loglik(ctx, k, 1, n, dx, 0, dy, 0, dv, 0, sigma, dval, g ? dg : (H ? dg : nullptr), 0, dH, n, 0, dc, nullptr));
    VSLAM_TRY(d2h(ctx, value, dval, 8));
    if (g) VSLAM_TRY(d2h(ctx, g, dg, n * 8));
    if (code) VSLAM_TRY(d2h(ctx, code, dc, L * 4));
    if (H) VSLAM_TRY(d2h_2d(ctx, H, (size_t)ldH * 8, dH, (size_t)n * 8, (size_t)n * 8, n));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}
