// SGRAF local similarity nodes, fused (EncoderSimilarity.forward, Fusionmodule.py:425-427):
//
//     Context_img = l2norm(P V)                       SCAN_attention's weighted context (:657-662)
//     sim_loc     = l2norm(W_loc (Context_img - cap_i)^2 + b_loc)
//
// One workgroup = one (image, 64-column caption tile) = 64 node rows, all S = 256 output features.
// The D-long squared-difference rows are never written to HBM: per 32-wide slice of D a wave
//   stage 1:  ctx[16 rows x 32] = P'[16 x 36] V[36 x 32]      (v_mfma_f32_16x16x4_f32, 18 per slice; P' = P / (||ctx|| + eps)
//             lives in registers for the whole kernel)
//             a = (ctx - E)^2  ->  LDS, directly in the A-operand layout of stage 2
//   stage 2:  acc[64 x 64 per wave] += a[64 x 32] W_loc[256 x 32]^T   (v_mfma_f32_32x32x2_f32, 64 per slice)
// so the matrix core spends 1 / 8 of its time on producing the operand, against an HBM round trip of 2 x 4 D bytes per
// node row before (270 MB written and read back per image at COCO size).  The epilogue adds the bias, l2-normalises the
// 256-wide rows (cross-wave reduction through LDS) and stores the nodes.
//
// LDS: A slice 2 x 8 KB + W_loc slice 2 x 32 KB = 80 KB -> two workgroups per CU.  Operand layout as in gemm_f32.hip:
// 8 planes of float4 (plane p = k / 4), physical row = row ^ p (conflict-free ds_write_b128 / ds_read_b128); the stage-1
// results are scattered with ds_write_b32 whose 32 lanes per cycle cover 32 distinct banks (4 planes x 8 rows).
#include "scan_common.h"

namespace itr {

constexpr int LM = 64, LN = 256, LK = 32, LPL = LK / 4;

struct LocArgs {
    const float *P;        // [nb][ncols][36]   attention weights (SCAN emit)
    const float *cn;       // [nb][ncols]       1 / (||ctx|| + eps)
    const float *img;      // [nb][36][D]       region embeddings of this image block
    const float *wtiled;   // [ncols][D]        tile-packed word embeddings
    const float *W;        // [256][D]          sim_tranloc_w.weight
    const float *bias;     // [256]
    float *X;              // [nb][ncols][256]  out: l2-normalised local nodes
    int64_t nb, n_tiles;
    int D;
};

struct LocSmem {
    float4 a[2][LPL][LM];   // 16 KB
    float4 w[2][LPL][LN];   // 64 KB
};

__global__ __launch_bounds__(256, 2) void sgraf_loc_kernel(LocArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LocSmem &sm = *reinterpret_cast<LocSmem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD x (blockIdx % 8) owns the caption tiles ct = x (mod 8) and walks the images of the block for each of them,
    // so a tile of word embeddings is pulled into one L2 only; W_loc (1 MB) is resident in every L2.
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, idx = bid >> 3;
    const int64_t ct = (idx / g.nb) * 8 + xcd, ii = idx % g.nb;
    if (ct >= g.n_tiles) return;
    const int64_t ncols = g.n_tiles * SC_NT;
    const int64_t row0 = ii * ncols + ct * SC_NT;
    const int D = g.D;
    const int nk = D / LK;

    // ---- stage-1 operands
    const int si = lane & 15, sg = lane >> 4;
    float pa[9];
    {
        const int64_t r = row0 + 16 * wave + si;
        const float sc = g.cn[r];
#pragma unroll
        for (int q = 0; q < 9; ++q) pa[q] = g.P[r * SC_R + 4 * q + sg] * sc;
    }
    // Loop-invariant addressing (the fp32 MFMA shares the vector ALU: the compiler's per-load 64-bit address arithmetic was
    // 225 VALU instructions per slice, ~19 % of the matrix time): uniform 64-bit bases advanced by scalar adds + ONE fixed
    // 32-bit per-lane byte offset per operand, loads issued through inline asm with our own s_waitcnt.
    const char *vbase = reinterpret_cast<const char *>(g.img + ii * SC_R * (int64_t)D);                       // V[k][d]
    const char *zbase = reinterpret_cast<const char *>(g.wtiled + (ct * SC_NT + 16 * wave) * (int64_t)D);    // E rows of this wave
    const char *wbase = reinterpret_cast<const char *>(g.W);
    const unsigned rowb = (unsigned)D * 4u;
    const unsigned voff_v = (unsigned)sg * rowb + si * 4u;            // V[4q + sg][d0 + si]      (q via the base, +16 columns via offset:64)
    const unsigned voff_z = 4u * sg * rowb + si * 4u;                 // E[4 sg + j][d0 + si]     (j via the base)
    float vb[2][9], zz[2][4];
#define LOC_LDG1(dst, base, voff, IMM) asm volatile("global_load_dword %0, %1, %2 offset:" #IMM : "=v"(dst) : "v"(voff), "s"(base) : "memory");
#define LOC_LDG4(dst, base, voff) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
#define LOC_LOAD_VZ(kc_)                                                                                        \
    {                                                                                                           \
        const char *vb_ = vbase + (int64_t)(kc_) * (LK * 4);                                                    \
        const char *zb_ = zbase + (int64_t)(kc_) * (LK * 4);                                                    \
        _Pragma("unroll") for (int q = 0; q < 9; ++q) {                                                         \
            const char *b_ = vb_ + (int64_t)(4 * q) * rowb;                                                     \
            LOC_LDG1(vb[0][q], b_, voff_v, 0) LOC_LDG1(vb[1][q], b_, voff_v, 64)                                \
        }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const char *b_ = zb_ + (int64_t)j * rowb;                                                           \
            LOC_LDG1(zz[0][j], b_, voff_z, 0) LOC_LDG1(zz[1][j], b_, voff_z, 64)                                \
        }                                                                                                       \
    }
#define LOC_STAGE1(buf_)                                                                                        \
    {                                                                                                           \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                      \
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};                                                                \
            _Pragma("unroll") for (int q = 0; q < 9; ++q)                                                       \
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[q], vb[nt][q], c, 0, 0, 0);                         \
            const int k = nt * 16 + si, p = k >> 2;                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                const float v = c[j] - zz[nt][j];                                                               \
                const int row = 16 * wave + 4 * sg + j;                                                         \
                reinterpret_cast<float *>(&sm.a[buf_][p][row ^ p])[k & 3] = v * v;                              \
            }                                                                                                   \
        }                                                                                                       \
    }

    // ---- stage-2 operands
    const int ld_row = tid >> 3, ld_p = tid & 7;
    f32x4 rw[8];
    const unsigned voff_w = (unsigned)ld_row * rowb + ld_p * 16u;     // W[ld_row + 32 s][kc*32 + 4 ld_p ..]  (s via the base)
#define LOC_GLOAD_W(kc_)                                                                                        \
    {                                                                                                           \
        const char *wb_ = wbase + (int64_t)(kc_) * (LK * 4);                                                    \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) { const char *b_ = wb_ + (int64_t)(32 * s) * rowb; LOC_LDG4(rw[s], b_, voff_w) } \
    }
// every asm load above has landed; the statements name the destination registers so that their consumers stay behind
#define LOC_VMWAIT                                                                                              \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rw[0]), "+v"(rw[1]), "+v"(rw[2]), "+v"(rw[3]), "+v"(rw[4]), "+v"(rw[5]), "+v"(rw[6]), "+v"(rw[7]), \
                 "+v"(zz[0][0]), "+v"(zz[0][1]), "+v"(zz[0][2]), "+v"(zz[0][3]), "+v"(zz[1][0]), "+v"(zz[1][1]), "+v"(zz[1][2]), "+v"(zz[1][3])::"memory"); \
    asm volatile("" : "+v"(vb[0][0]), "+v"(vb[0][1]), "+v"(vb[0][2]), "+v"(vb[0][3]), "+v"(vb[0][4]), "+v"(vb[0][5]), "+v"(vb[0][6]), "+v"(vb[0][7]), \
                 "+v"(vb[0][8]), "+v"(vb[1][0]), "+v"(vb[1][1]), "+v"(vb[1][2]), "+v"(vb[1][3]), "+v"(vb[1][4]), "+v"(vb[1][5]), "+v"(vb[1][6]), \
                 "+v"(vb[1][7]), "+v"(vb[1][8])::"memory");
#define LOC_LSTORE_W(buf_)                                                                                      \
    { _Pragma("unroll") for (int s = 0; s < 8; ++s) *reinterpret_cast<f32x4 *>(&sm.w[buf_][ld_p][(ld_row + 32 * s) ^ ld_p]) = rw[s]; }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LOC_GLOAD_W(0)
    LOC_LOAD_VZ(0)
    LOC_VMWAIT
    LOC_STAGE1(0)
    LOC_LSTORE_W(0)
    __syncthreads();

    const int fi = lane & 31, fg = lane >> 5;
    // Branch-free K loop (one basic block per slice, so that the scheduler may overlap the phases): the tail iteration
    // re-produces the last slice into the idle buffer.  Issue order requested from the scheduler: the global loads of
    // the next slice go out first, one per stage-2 MFMA; then the fragment reads; the stage-1 MFMAs, the squared
    // difference and the LDS stores of the next slice close the iteration.
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        const int kn = kc + 1 < nk ? kc + 1 : nk - 1;
        LOC_GLOAD_W(kn)
        LOC_LOAD_VZ(kn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = 2 * q + fg;
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sm.a[buf][p][(i * 32 + fi) ^ p];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sm.w[buf][p][(wave * 64 + j * 32 + fi) ^ p];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
        LOC_VMWAIT
        LOC_STAGE1(buf ^ 1)
        LOC_LSTORE_W(buf ^ 1)
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // first fragment reads
#pragma unroll
        for (int i_ = 0; i_ < 34; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 // VMEM read
        }
#pragma unroll
        for (int i_ = 0; i_ < 12; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 // DS read
        }
        __syncthreads();
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef LOC_LDG1
#undef LOC_LDG4
#undef LOC_VMWAIT
#undef LOC_LOAD_VZ
#undef LOC_STAGE1
#undef LOC_GLOAD_W
#undef LOC_LSTORE_W

    // ---- epilogue: + bias, l2norm over the 256 features of a row (utils.py:10-15, eps 1e-8), store
    // acc[i][j][r]: row = i*32 + (r & 3) + 8 (r >> 2) + 4 fg, column = wave*64 + j*32 + fi
    float ss[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bv = g.bias[wave * 64 + j * 32 + fi];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = acc[i][0][r] * acc[i][0][r] + acc[i][1][r] * acc[i][1][r];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            ss[i][r] = s;
        }
    __syncthreads();                                    // every wave is done with the operand buffers
    float *part = reinterpret_cast<float *>(smem_raw);  // [4 waves][64 rows]
    if (fi == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave * LM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg] = ss[i][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
            const float tot = part[row] + part[LM + row] + part[2 * LM + row] + part[3 * LM + row];
            const float rn = 1.f / (sqrtf(tot) + 1e-8f);
            float *out = g.X + (row0 + row) * LN + wave * 64 + fi;
            out[0] = acc[i][0][r] * rn;
            out[32] = acc[i][1][r] * rn;
        }
}

int sgraf_loc_fused(const float *P, const float *cn, const float *img, const float *wtiled, const float *W, const float *bias,
                    float *X, int64_t nb, int64_t n_tiles, int D, hipStream_t st) {
    static_assert(sizeof(LocSmem) == 80 * 1024, "two workgroups per CU");
    static bool attr_done = false;   // idempotent: racing callers set the same value
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sgraf_loc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(LocSmem));
        if (e != hipSuccess) {
            set_error("sgraf_loc: cannot reserve %zu B of LDS: %s", sizeof(LocSmem), hipGetErrorString(e));
            return ITR_ERR_HIP;
        }
        attr_done = true;
    }
    if (nb == 0 || n_tiles == 0) return ITR_OK;
    LocArgs g{P, cn, img, wtiled, W, bias, X, nb, n_tiles, D};
    const int64_t grid = ceil_div(n_tiles, (int64_t)8) * 8 * nb;
    hipLaunchKernelGGL(sgraf_loc_kernel, dim3((unsigned)grid), dim3(256), sizeof(LocSmem), st, g);
    ITR_CHECK_LAUNCH("sgraf_loc");
    return ITR_OK;
}

}  // namespace itr
