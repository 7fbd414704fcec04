// Micro-benchmark for VERDICT r5 #4: the ATTENTION phases of the fused SGR kernel (csrc/sgr_fused.hip P2: E = Q' X^T, softmax,
// Y = P X per graph; Fusionmodule.py:564-587) in two tilings, operands in LDS exactly as the kernel holds them (two [64][260] fp32
// node-row buffers, graphs = runs of rows):
//   form A (today): v_mfma_f32_16x16x4_f32, one wave per (graph, 16-query tile), E^T and the softmax in the accumulator layout,
//                   graphs of 17-21 nodes padded to 32 key rows / two query tiles;
//   form B        : v_mfma_f32_4x4x1_16B_f32 (16 independent 4 x 4 x 1 blocks per instruction, the same flop rate), graphs padded
//                   to a multiple of 4 rows: E as ceil(nb / (16 / nb)) instructions per k (key operand shared by the sets), the k
//                   range split over two waves and added through LDS, softmax from LDS, Y as (row block, 64-column set) units.
// Both forms run one barrier per phase boundary they need; the result of both is checked against a float64 host reference.
// Output: cycles (s_memtime) per P2 of a 64-row group, per graph size.   hipcc --offload-arch=gfx950 -O3 -o sgr_attn_blocks sgr_attn_blocks.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 260, ROWS = 64, S = 256, THREADS = 512, MAXG = 8;

struct Group { int ng; int base[MAXG]; int n[MAXG]; };

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void load_rows(float *X, float *Q, const float *Xg, const float *Qg, int tid) {
    for (int i = tid; i < ROWS * S / 4; i += THREADS) {
        const int row = i / (S / 4), c4 = i % (S / 4);
        *reinterpret_cast<f32x4 *>(&X[row * LD + 4 * c4]) = reinterpret_cast<const f32x4 *>(Xg)[i];
        *reinterpret_cast<f32x4 *>(&Q[row * LD + 4 * c4]) = reinterpret_cast<const f32x4 *>(Qg)[i];
    }
}
__device__ __forceinline__ void store_rows(const float *Q, float *Yg, int tid) {
    for (int i = tid; i < ROWS * S / 4; i += THREADS) {
        const int row = i / (S / 4), c4 = i % (S / 4);
        reinterpret_cast<f32x4 *>(Yg)[i] = *reinterpret_cast<const f32x4 *>(&Q[row * LD + 4 * c4]);
    }
}

// ---------------------------------------------------------------- form A: 16 x 16 x 4 tiles, one wave per (graph, query tile)
__global__ __launch_bounds__(THREADS) void form_a(Group g, const float *Xg, const float *Qg, float *Yg, int iters, long long *cyc, long long *ph) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *X = lds, *Q = lds + ROWS * LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fi = lane & 15, fq = lane >> 4;
    load_rows(X, Q, Xg, Qg, tid);
    int ug = -1, ut = 0;
    for (int gi = 0, u = 0; gi < g.ng; ++gi)
        for (int t = 0; t < (g.n[gi] + 15) / 16; ++t, ++u)
            if (u == wave) { ug = gi; ut = t; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long pe = 0, ps = 0, py = 0, pb = 0;        // wave 0's phases (unit 0 = the first graph's first query tile)
    for (int it = 0; it < iters; ++it) {
        const long long s0 = __builtin_amdgcn_s_memtime();
        long long s1 = s0, s2 = s0, s3 = s0;
        if (ug >= 0) {
            const int n = g.n[ug], base = g.base[ug], nkt = (n + 15) / 16;
            f32x4 e[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            const float *qp = &Q[(base + min(16 * ut + fi, n - 1)) * LD + 4 * fq];
            const float *x0 = &X[(base + min(fi, n - 1)) * LD + 4 * fq];
            const float *x1 = &X[(base + min(16 + fi, n - 1)) * LD + 4 * fq];
            if (nkt == 2) {
#pragma unroll 4
                for (int kb = 0; kb < S / 16; ++kb) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(qp + 16 * kb);
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(x0 + 16 * kb), a1 = *reinterpret_cast<const f32x4 *>(x1 + 16 * kb);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { e[0] = MFMA16(a0[c], b[c], e[0]); e[1] = MFMA16(a1[c], b[c], e[1]); }
                }
            } else {
#pragma unroll 4
                for (int kb = 0; kb < S / 16; ++kb) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(qp + 16 * kb);
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(x0 + 16 * kb);
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[0] = MFMA16(a0[c], b[c], e[0]);
                }
            }
            // softmax over the keys j = 16 kt + 4 fq + r of query column fi, in the accumulator layout
            asm volatile("s_nop 0" : "+v"(e[0]), "+v"(e[1]));
            s1 = __builtin_amdgcn_s_memtime();
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt >= nkt || 16 * kt + 4 * fq + r >= n) e[kt][r] = -INFINITY;
                    m = fmaxf(m, e[kt][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[kt][r] = __expf(e[kt][r] - m); sum += e[kt][r]; }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) e[kt][r] *= inv;
            // Y^T[d, i] = sum_j X[j, d] P^T[j, i]
            asm volatile("s_nop 0" : "+v"(e[0]), "+v"(e[1]));
            s2 = __builtin_amdgcn_s_memtime();
            int xrow[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xrow[kt][r] = (base + min(16 * kt + 4 * fq + r, n - 1)) * LD + fi;
            const bool live = 16 * ut + fi < n;
            float *yq = &Q[(base + 16 * ut + fi) * LD + 4 * fq];
#pragma unroll 4
            for (int dt = 0; dt < S / 16; ++dt) {
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = MFMA16(X[xrow[0][r] + 16 * dt], e[0][r], acc);
                if (nkt == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = MFMA16(X[xrow[1][r] + 16 * dt], e[1][r], acc);
                }
                if (live) *reinterpret_cast<f32x4 *>(yq + 16 * dt) = acc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            s3 = __builtin_amdgcn_s_memtime();
        }
        __syncthreads();
        pe += s1 - s0; ps += s2 - s1; py += s3 - s2; pb += __builtin_amdgcn_s_memtime() - s3;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; ph[blockIdx.x * 4] = pe; ph[blockIdx.x * 4 + 1] = ps; ph[blockIdx.x * 4 + 2] = py; ph[blockIdx.x * 4 + 3] = pb; }
    if (blockIdx.x == 0 && Yg) store_rows(Q, Yg, tid);
}

// ---------------------------------------------------------------- form B: 16 4 x 4 x 1 blocks per instruction
// LDS after the two row buffers: EP[2][72][25] (the two k halves of E), P[72][28].
constexpr int EP_LD = 25, P_LD = 28, PADROWS = 72 + 24;
// Y[row 4 bi + i][64 c + 4 blk + j] for a graph of NB row blocks: the 16 blocks of an instruction are 16 column blocks of one row block;
// the wave keeps the 4 NB key rows of its 64 columns in registers across the row blocks.  (Templated: with nb a run-time value hipcc
// wrapped every load in its own exec-mask branch -- the first version of this file.)
template <int NB>
__device__ __forceinline__ void b_y(const float *X, float *Q, const float *P, int n, int base, int eoff, int h, int wpg, int lane, int li) {
    for (int c = h * (4 / wpg); c < (h + 1) * (4 / wpg); ++c) {
        float b[4 * NB];
#pragma unroll
        for (int k = 0; k < 4 * NB; ++k) b[k] = X[(base + min(k, n - 1)) * LD + 64 * c + lane];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            const float *pp = &P[(eoff + 4 * bi + li) * P_LD];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(pp + 4 * q);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc[cc] = MFMA4(a[cc], b[4 * q + cc], acc[cc]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (bi < NB - 1 || 4 * bi + i < n) Q[(base + 4 * bi + i) * LD + 64 * c + lane] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        }
    }
}
__global__ __launch_bounds__(THREADS) void form_b(Group g, const float *Xg, const float *Qg, float *Yg, int iters, long long *cyc, long long *ph) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *X = lds, *Q = lds + ROWS * LD, *EP = Q + ROWS * LD, *P = EP + 2 * PADROWS * EP_LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, blk = lane >> 2, li = lane & 3;
    load_rows(X, Q, Xg, Qg, tid);
    for (int i = tid; i < PADROWS * P_LD; i += THREADS) P[i] = 0.f;
    for (int i = tid; i < 2 * PADROWS * EP_LD; i += THREADS) EP[i] = 0.f;
    const int wpg = g.ng <= 4 ? 2 : 1;                  // waves per graph
    const int gi = wave / wpg, h = wave % wpg;
    const bool on = gi < g.ng;
    int n = 1, base = 0, eoff = 0;
    if (on) { n = g.n[gi]; base = g.base[gi]; for (int q = 0; q < gi; ++q) eoff += 4 * ((g.n[q] + 3) / 4); }
    const int nb = (n + 3) / 4, rps = min(16 / nb, nb), nsets = (nb + rps - 1) / rps, kpad = 4 * nb;
    const int bl = blk / nb, bj = blk % nb;
    const bool valid = bl < rps;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long pe = 0, ps = 0, py = 0, pb = 0;        // wave 0's phases; pb = the three barriers' waits
    for (int it = 0; it < iters; ++it) {
        const long long s0 = __builtin_amdgcn_s_memtime();
        if (on) {
            // E[query 4 bi + i][key 4 bj + j]: blocks (bi, bj); set s holds bi = s rps .. s rps + rps - 1, every bj
            f32x4 acc[3][4];            // four independent chains per set (a dependent 4x4x1 chain is latency-bound: first version of this file)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[s][c] = f32x4{0, 0, 0, 0};
            const int k0 = wpg == 2 ? 128 * h : 0, nk4 = wpg == 2 ? 32 : 64;
            const float *kp = &X[(base + min(4 * bj + li, n - 1)) * LD + k0];
            const float *q0 = &Q[(base + min(4 * bl + li, n - 1)) * LD + k0];
            const float *q1 = &Q[(base + min(4 * (rps + bl) + li, n - 1)) * LD + k0];
            const float *q2 = &Q[(base + min(4 * (2 * rps + bl) + li, n - 1)) * LD + k0];
            if (nsets == 1) {
#pragma unroll 4
                for (int k4 = 0; k4 < nk4; ++k4) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(kp + 4 * k4), a0 = *reinterpret_cast<const f32x4 *>(q0 + 4 * k4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[0][c] = MFMA4(a0[c], b[c], acc[0][c]);
                }
            } else if (nsets == 2) {
#pragma unroll 4
                for (int k4 = 0; k4 < nk4; ++k4) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(kp + 4 * k4), a0 = *reinterpret_cast<const f32x4 *>(q0 + 4 * k4),
                                a1 = *reinterpret_cast<const f32x4 *>(q1 + 4 * k4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { acc[0][c] = MFMA4(a0[c], b[c], acc[0][c]); acc[1][c] = MFMA4(a1[c], b[c], acc[1][c]); }
                }
            } else {
#pragma unroll 4
                for (int k4 = 0; k4 < nk4; ++k4) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(kp + 4 * k4), a0 = *reinterpret_cast<const f32x4 *>(q0 + 4 * k4),
                                a1 = *reinterpret_cast<const f32x4 *>(q1 + 4 * k4), a2 = *reinterpret_cast<const f32x4 *>(q2 + 4 * k4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { acc[0][c] = MFMA4(a0[c], b[c], acc[0][c]); acc[1][c] = MFMA4(a1[c], b[c], acc[1][c]); acc[2][c] = MFMA4(a2[c], b[c], acc[2][c]); }
                }
            }
            if (valid) {
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (s < nsets) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = 4 * (s * rps + bl) + i;
                            if (row < kpad) EP[(h * PADROWS + eoff + row) * EP_LD + 4 * bj + li] = (acc[s][0][i] + acc[s][1][i]) + (acc[s][2][i] + acc[s][3][i]);
                        }
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long s1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const long long s2 = __builtin_amdgcn_s_memtime();
        if (on) {
            // softmax rows from LDS: one lane per row (the graph's rows dealt to its waves), the row's <= 24 scores in registers
            const int r = wpg * lane + h;
            if (r < n) {
                float v[24], m = -INFINITY, sum = 0.f;
#pragma unroll
                for (int col = 0; col < 24; ++col) {
                    const float t = EP[(eoff + r) * EP_LD + col] + EP[(PADROWS + eoff + r) * EP_LD + col];      // (the second half is zero when one wave ran all of k)
                    v[col] = col < n ? t : -INFINITY;
                    m = fmaxf(m, v[col]);
                }
#pragma unroll
                for (int col = 0; col < 24; ++col) { v[col] = __expf(v[col] - m); sum += v[col]; }
                const float inv = 1.f / sum;
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (q < nb) *reinterpret_cast<f32x4 *>(&P[(eoff + r) * P_LD + 4 * q]) = f32x4{v[4 * q] * inv, v[4 * q + 1] * inv, v[4 * q + 2] * inv, v[4 * q + 3] * inv};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long s3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const long long s4 = __builtin_amdgcn_s_memtime();
        if (on) switch (nb) {
            case 1: b_y<1>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
            case 2: b_y<2>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
            case 3: b_y<3>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
            case 4: b_y<4>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
            case 5: b_y<5>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
            default: b_y<6>(X, Q, P, n, base, eoff, h, wpg, lane, li); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long s5 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        pe += s1 - s0; ps += s3 - s2; py += s5 - s4; pb += (s2 - s1) + (s4 - s3) + (__builtin_amdgcn_s_memtime() - s5);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; ph[blockIdx.x * 4] = pe; ph[blockIdx.x * 4 + 1] = ps; ph[blockIdx.x * 4 + 2] = py; ph[blockIdx.x * 4 + 3] = pb; }
    if (blockIdx.x == 0 && Yg) store_rows(Q, Yg, tid);
}

// which (lane, register) of D a one-hot A lane x one-hot B lane lands in: documents the 4x4x1 operand map the kernel assumes
__global__ void probe(int la, int lb, float *out) {
    const int lane = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = MFMA4(lane == la ? 1.f : 0.f, lane == lb ? 1.f : 0.f, acc);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void reference(const Group &g, const std::vector<float> &X, const std::vector<float> &Q, std::vector<double> &Y) {
    Y.assign((size_t)ROWS * S, 0.0);
    for (int gi = 0; gi < g.ng; ++gi) {
        const int n = g.n[gi], b = g.base[gi];
        for (int i = 0; i < n; ++i) {
            std::vector<double> e(n);
            double m = -1e300, sum = 0;
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int k = 0; k < S; ++k) s += (double)Q[(b + i) * S + k] * X[(b + j) * S + k];
                e[j] = s; m = fmax(m, s);
            }
            for (int j = 0; j < n; ++j) { e[j] = exp(e[j] - m); sum += e[j]; }
            for (int d = 0; d < S; ++d) {
                double y = 0;
                for (int j = 0; j < n; ++j) y += e[j] / sum * X[(b + j) * S + d];
                Y[(size_t)(b + i) * S + d] = y;
            }
        }
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz (s_memtime counts at 100 MHz on gfx9: cycles below = event time x nominal clock)\n", prop.name, cus, prop.clockRate / 1000);
    {   // operand map of v_mfma_f32_4x4x1_16B_f32
        float *d; CK(hipMalloc(&d, 256 * 4));
        float hst[256];
        const int probes[3][2] = {{0, 0}, {5, 6}, {62, 61}};
        for (auto &p : probes) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p[0], p[1], d);
            CK(hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
                if (hst[l * 4 + r] != 0.f) printf("# probe A lane %d x B lane %d -> D register %d of lane %d\n", p[0], p[1], r, l);
        }
        CK(hipFree(d));
    }
    const size_t lds_a = 2 * (size_t)ROWS * LD * 4, lds_b = lds_a + (2 * (size_t)PADROWS * EP_LD + (size_t)PADROWS * P_LD) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(form_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(form_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    std::vector<float> X((size_t)ROWS * S), Q((size_t)ROWS * S);
    srand(7);
    for (auto &v : X) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    for (auto &v : Q) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    float *dX, *dQ, *dY; long long *dc, *dph;
    const int grid = cus * 2, iters = 400;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dY, X.size() * 4)); CK(hipMalloc(&dc, grid * 8)); CK(hipMalloc(&dph, grid * 32));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    struct Case { const char *name; std::vector<int> n; };
    std::vector<Case> cases = {{"7 x 9", {9, 9, 9, 9, 9, 9, 9}}, {"5 x 12", {12, 12, 12, 12, 12}}, {"4 x 13", {13, 13, 13, 13}}, {"4 x 16", {16, 16, 16, 16}},
                               {"3 x 17", {17, 17, 17}}, {"3 x 19", {19, 19, 19}}, {"3 x 20", {20, 20, 20}}, {"3 x 21", {21, 21, 21}},
                               {"21 17 13 13", {21, 17, 13, 13}}, {"20 16 12 9 7", {20, 16, 12, 9, 7}}, {"21 21 14 8", {21, 21, 14, 8}}};
    printf("%-16s %6s %6s | %12s %12s %7s | %10s %10s | %9s %9s\n", "graphs (nodes)", "rows", "units", "A cyc/P2", "B cyc/P2", "A/B", "A max err", "B max err", "A memtime", "B memtime");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &c : cases) {
        Group g{}; g.ng = (int)c.n.size();
        int rows = 0, units = 0;
        for (int i = 0; i < g.ng; ++i) { g.base[i] = rows; g.n[i] = c.n[i]; rows += c.n[i]; units += (c.n[i] + 15) / 16; }
        if (rows > ROWS || units > 8 || g.ng > MAXG) { printf("%-16s skipped\n", c.name); continue; }
        std::vector<double> ref; reference(g, X, Q, ref);
        double err[2], cyc[2], tick[2], phs[2][4];
        std::vector<float> Y(X.size());
        for (int f = 0; f < 2; ++f) {
            auto launch = [&](int it, float *y) {
                if (f == 0) hipLaunchKernelGGL(form_a, dim3(grid), dim3(THREADS), lds_a, 0, g, dX, dQ, y, it, dc, dph);
                else hipLaunchKernelGGL(form_b, dim3(grid), dim3(THREADS), lds_b, 0, g, dX, dQ, y, it, dc, dph);
            };
            launch(1, dY); CK(hipDeviceSynchronize());
            CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
            double me = 0;
            for (int gi = 0; gi < g.ng; ++gi)
                for (int r = g.base[gi]; r < g.base[gi] + g.n[gi]; ++r)
                    for (int d = 0; d < S; ++d) me = fmax(me, fabs(Y[(size_t)r * S + d] - ref[(size_t)r * S + d]));
            err[f] = me;
            launch(iters, nullptr); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(iters, nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // two workgroups per CU run one after the other (LDS): time / (2 iters) = one P2
            cyc[f] = ms * 1e-3 * (prop.clockRate * 1e3) / (2.0 * iters);
            std::vector<long long> hc(grid);
            CK(hipMemcpy(hc.data(), dc, grid * 8, hipMemcpyDeviceToHost));
            double tot = 0;
            for (auto v : hc) tot += (double)v;
            tick[f] = tot / grid / iters;
            std::vector<long long> hp(grid * 4);
            CK(hipMemcpy(hp.data(), dph, grid * 32, hipMemcpyDeviceToHost));
            for (int q = 0; q < 4; ++q) { double t = 0; for (int w = 0; w < grid; ++w) t += (double)hp[w * 4 + q]; phs[f][q] = t / grid / iters; }
        }
        printf("%-16s %6d %6d | %12.0f %12.0f %7.2f | %10.2e %10.2e | %9.1f %9.1f\n", c.name, rows, units, cyc[0], cyc[1], cyc[0] / cyc[1], err[0], err[1], tick[0], tick[1]);
        printf("    wave 0 (first graph), s_memtime per P2:  A: E %.0f  softmax %.0f  Y %.0f  barrier %.0f   |   B: E %.0f  softmax %.0f  Y %.0f  barriers %.0f\n",
               phs[0][0], phs[0][1], phs[0][2], phs[0][3], phs[1][0], phs[1][1], phs[1][2], phs[1][3]);
    }
    return 0;
}
